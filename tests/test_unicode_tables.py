"""`not gpu`: the generated Unicode class tables against `unicodedata` 13.0, recomputed HERE independently of
tools/gen_unicode_tables.py.  Both copies (the oracle's and the product's) come out of that one generator, so a wrong class
would be common-mode and invisible to every oracle-vs-device test: this test is what would catch it."""
import os
import re
import unicodedata

import pytest

from conftest import ROOT

COPIES = {"oracle": os.path.join(ROOT, "oracle"), "product": os.path.join(ROOT, "tokenizer_amd", "csrc")}
NAMES = {0: "OTHER", 1: "Lu", 2: "Ll", 3: "Lt", 4: "Lm", 5: "Lo", 6: "M", 7: "N", 8: "WS"}


def expected_class(cp):
    """What the patterns' classes mean: \\p{L} subcategories, \\p{M} = Mn|Mc|Me, \\p{N} = Nd|Nl|No,
    \\s (.NET char.IsWhiteSpace) = Zs|Zl|Zp + U+0009..U+000D + U+0085.  Surrogate code units are Cs: OTHER."""
    if 0xD800 <= cp <= 0xDFFF:
        return 0
    cat = unicodedata.category(chr(cp))
    if cat in ("Zs", "Zl", "Zp") or 0x09 <= cp <= 0x0D or cp == 0x85:
        return 8
    if cat[0] == "L":
        return {"Lu": 1, "Ll": 2, "Lt": 3, "Lm": 4, "Lo": 5}[cat]
    if cat[0] == "M":
        return 6
    if cat[0] == "N":
        return 7
    return 0


def read_ranges(path):
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    return [(int(a, 16), int(b, 16), int(c)) for a, b, c in re.findall(r"\{0x([0-9A-Fa-f]+),0x([0-9A-Fa-f]+),(\d+)\}", text)]


@pytest.mark.parametrize("copy", sorted(COPIES))
def test_bmp_classes_match_unicodedata(copy):
    assert unicodedata.unidata_version == "13.0.0"           # net6.0's Unicode data
    table = [0] * 0x10000
    prev = -1
    for a, b, c in read_ranges(os.path.join(COPIES[copy], "unicode13_classes.inc")):
        assert prev < a <= b <= 0xFFFF and 1 <= c <= 8
        prev = b
        for u in range(a, b + 1):
            table[u] = c
    bad = [(hex(u), NAMES[table[u]], NAMES[expected_class(u)]) for u in range(0x10000) if table[u] != expected_class(u)]
    assert not bad, bad[:10]
    # a few anchors written out by hand
    for u, c in ((0x41, 1), (0x61, 2), (0x1C5, 3), (0x2B0, 4), (0x4E2D, 5), (0x301, 6), (0x39, 7), (0x216B, 7), (0xBD, 7), (0x660, 7),
                 (0x20, 8), (0x85, 8), (0xA0, 8), (0x3000, 8), (0x2028, 8), (0x1C, 0), (0xFEFF, 0), (0x200D, 0), (0x2B50, 0), (0xD83D, 0), (0x17F, 2), (0x212A, 1)):
        assert table[u] == c, hex(u)


@pytest.mark.parametrize("copy", sorted(COPIES))
def test_supplementary_classes_match_unicodedata(copy):
    rs = read_ranges(os.path.join(COPIES[copy], "unicode13_supp.inc"))
    prev = 0xFFFF
    covered = {}
    for a, b, c in rs:
        assert prev < a <= b <= 0x10FFFF and 1 <= c <= 7     # (there is no white space above the BMP)
        prev = b
        for u in range(a, b + 1):
            covered[u] = c
    bad = [hex(u) for u in range(0x10000, 0x110000) if covered.get(u, 0) != expected_class(u)]
    assert not bad, bad[:10]
    for u, c in ((0x20000, 5), (0x1D7CF, 7), (0x1D400, 1), (0x10428, 2), (0x1F600, 0), (0x1D165, 6), (0x10107, 7), (0xE0100, 6)):
        assert covered.get(u, 0) == c, hex(u)


def test_the_two_copies_are_identical():
    for name in ("unicode13_classes.inc", "unicode13_supp.inc"):
        assert open(os.path.join(COPIES["oracle"], name)).read() == open(os.path.join(COPIES["product"], name)).read()


def test_device_table_image_matches_unicodedata():
    """The table the product uploads (one byte per code point below 0x40000 + the ranges above), read back through the C ABI:
    all 1,114,112 code points against unicodedata, so the folding of the ranges into the direct table is checked too."""
    import ctypes as C

    import numpy as np

    from tokenizer_amd import _native as N
    L = N.default_library().L
    got = np.zeros(0x110000, np.uint8)
    L.tkz_unicode_classes(0, 0x110000, got.ctypes.data_as(C.c_void_p))
    exp = np.fromiter((expected_class(u) for u in range(0x110000)), np.uint8, 0x110000)
    bad = np.nonzero(got != exp)[0]
    assert len(bad) == 0, [(hex(int(u)), int(got[u]), int(exp[u])) for u in bad[:10]]
