#!/usr/bin/env python3
"""Emit tokenizer_amd/csrc/tkz_corpus_words.inc: the fixed word table of the synthetic corpus
generator (tkz_corpus.h).  4096 entries of up to 23 ASCII letters, most frequent first; the generator
draws ranks log-uniformly (Zipf s = 1).  The head is hand-written common English, the tail
is derivational morphology (prefix + stem + suffix) sorted by length, so that the tail is made of long,
rare words: about 5 % of the table is longer than 16 bytes (`internationalizations`, ...), which is what
sends a piece to the general merge path of the encode kernel, and a realistic share of all pieces misses
the whole-piece lookup."""
import sys

COMMON = """
the of and to a in is that it was for on are with as his they be at one have this from or had by
not but what some we can out other were all there when up use your how said an each she which do
their time if will way about many then them write would like so these her long make thing see him
two has look more day could go come did number sound no most people my over know water than call
first who may down side been now find any new work part take get place made live where after back
little only round man year came show every good me give our under name very through just form
sentence great think say help low line differ turn cause much mean before move right boy old too
same tell does set three want air well also play small end put home read hand port large spell add
even land here must big high such follow act why ask men change went light kind off need house
picture try us again animal point mother world near build self earth father head stand own page
should country found answer school grow study still learn plant cover food sun four between state
keep eye never last let thought city tree cross farm hard start might story saw far sea draw left
late run while press close night real life few north open seem together next white children begin
got walk example ease paper group always music those both mark often letter until mile river car
feet care second book carry took science eat room friend began idea fish mountain stop once base
hear horse cut sure watch color face wood main enough plain girl usual young ready above ever red
list though feel talk bird soon body dog family direct pose leave song measure door product black
short numeral class wind question happen complete ship area half rock order fire south problem
piece told knew pass since top whole king space heard best hour better true during hundred five
remember step early hold west ground interest reach fast verb sing listen six table travel less
morning ten simple several vowel toward war lay against pattern slow center love person money serve
appear road map rain rule govern pull cold notice voice unit power town fine certain fly fall lead
cry dark machine note wait plan figure star box noun field rest correct able pound done beauty
drive stood contain front teach week final gave green oh quick develop ocean warm free minute
strong special mind behind clear tail produce fact street inch multiply nothing course stay wheel
full force blue object decide surface deep moon island foot system busy test record boat common
gold possible plane stead dry wonder laugh thousand ago ran check game shape equate hot miss
brought heat snow tire bring yes distant fill east paint language among grand ball yet wave drop
heart am present heavy dance engine position arm wide sail material size vary settle speak weight
general ice matter circle pair include divide syllable felt perhaps pick sudden count square
reason length represent art subject region energy hunt probable bed brother egg ride cell believe
fraction forest sit race window store summer train sleep prove lone leg exercise wall catch mount
wish sky board joy winter sat written wild instrument kept glass grass cow job edge sign visit
past soft fun bright gas weather month million bear finish happy hope flower clothe strange gone
jump baby eight village meet root buy raise solve metal whether push seven paragraph third shall
held hair describe cook floor either result burn hill safe cat century consider type law bit
coast copy phrase silent tall sand soil roll temperature finger industry value fight lie beat
excite natural view sense ear else quite broke case middle kill son lake moment scale loud spring
observe child straight consonant nation dictionary milk speed method organ pay age section dress
cloud surprise quiet stone tiny climb cool design poor lot experiment bottom key iron single stick
flat twenty skin smile crease hole trade melody trip office receive row mouth exact symbol die
least trouble shout except wrote seed tone join suggest clean break lady yard rise bad blow oil
blood touch grew cent mix team wire cost lost brown wear garden equal sent choose fell fit flow
fair bank collect save control decimal gentle woman captain practice separate difficult doctor
please protect noon whose locate ring character insect caught period indicate radio spoke atom
human history effect electric expect crop modern element hit student corner party supply bone
rail imagine provide agree thus capital chair danger fruit rich thick soldier process operate
guess necessary sharp wing create neighbor wash bat rather crowd corn compare poem string bell
depend meat rub tube famous dollar stream fear sight thin triangle planet hurry chief colony clock
mine tie enter major fresh search send yellow gun allow print dead spot desert suit current lift
rose continue block chart hat sell success company subtract event particular deal swim term
opposite wife shoe shoulder spread arrange camp invent cotton born determine quart nine truck
noise level chance gather shop stretch throw shine property column molecule select wrong gray
repeat require broad prepare salt nose plural anger claim continent oxygen sugar death pretty
skill women season solution magnet silver thank branch match suffix especially fig afraid huge
sister steel discuss forward similar guide experience score apple bought led pitch coat mass card
band rope slip win dream evening condition feed tool total basic smell valley nor double seat
arrive master track parent shore division sheet substance favor connect post spend chord fat glad
original share station dad bread charge proper bar offer segment slave duck instant market degree
populate chick dear enemy reply drink occur support speech nature range steam motion path liquid
log meant quotient teeth shell neck
""".split()

# stems the tail is derived from (technical + general); a stem ending in a vowel-dropping form is spelt as it combines
STEMS = """token encod comput network memor kernel vector matri parallel schedul compil optim gradient tensor buffer
thread latenc bandwidth quantiz normaliz national character organiz general special standard local global central
industrial commercial material natural formal legal social critical practical technical political historical
physical chemical logical mechanical electrical numerical statistical theoretical experimental environmental
professional institutional constitutional conventional functional operational educational international
sequenc structur configur architectur infrastructur represent implement document instrument environ govern
develop manag establish accomplish acknowledg understand communicat demonstrat investigat administrat
concentrat illustrat incorporat differentiat discriminat authenticat synchroniz initializ serializ virtualiz
visualiz categoriz characteriz internationaliz institutionaliz compartmentaliz conceptualiz
respons product construct instruct destruct abstract interact transact distribut contribut attribut
""".split()
SUFF = ["", "s", "ed", "er", "ers", "ing", "ings", "ation", "ations", "ized", "izing", "ize", "izes", "al", "ally", "ity", "ities",
        "able", "ability", "ly", "ment", "ments", "ness", "ism", "ist", "ists", "ive", "ively", "iveness", "or", "ors", "ional", "ionally"]
PREF = ["", "re", "un", "pre", "de", "non", "mis", "dis", "over", "under", "inter", "multi", "micro", "macro", "anti", "counter",
        "super", "trans", "hyper", "pseudo", "meta", "cross", "auto", "self", "semi", "sub", "post", "co", "out", "up"]
WIDTH = 24
N_WORDS = 4096


def main():
    seen, head = set(), []
    for w in COMMON:
        if w not in seen and w.isalpha() and len(w) < WIDTH:
            seen.add(w); head.append(w)
    # the tail: every prefix + stem + suffix, deterministically thinned, shortest first (rank grows with length, as in text)
    tail = []
    k = 0
    for pi, p in enumerate(PREF):
        for si, s in enumerate(STEMS):
            for ui, u in enumerate(SUFF):
                k += 1
                if (pi * 7 + si * 13 + ui * 29 + (k * 2654435761 >> 7)) % 11 >= 4:      # keep ~4 of 11 combinations
                    continue
                stem = s
                if u and u[0] in "aei" and stem.endswith("e"):
                    stem = stem[:-1]
                w = p + stem + u
                if w not in seen and len(w) < WIDTH and w.isalpha():
                    seen.add(w); tail.append(w)
    tail.sort(key=lambda w: (len(w), w))
    need = N_WORDS - len(head)
    # quotas by length class, so that the table's tail looks like a real lexicon's: mostly 5..12 letters, some 13..16, and
    # ~4 % of the table longer than 16 bytes -- all of those in the last Zipf octave (the rarest ranks)
    short = [w for w in tail if len(w) <= 12]
    mid = [w for w in tail if 13 <= len(w) <= 16]
    long_ = [w for w in tail if len(w) > 16]
    n_long, n_mid = 160, 640
    n_short = need - n_long - n_mid
    assert len(short) >= n_short and len(mid) >= n_mid and len(long_) >= n_long, (len(short), len(mid), len(long_))

    def spread(lst, n):
        step = len(lst) / float(n)
        return [lst[int(i * step)] for i in range(n)]
    picked = spread(short, n_short) + spread(mid, n_mid) + spread(long_, n_long)
    words = head + picked
    assert len(words) == N_WORDS and len(set(words)) == N_WORDS
    out = ["/* GENERATED by tools/gen_corpus_words.py -- %d words, %d bytes each, NUL padded */\n" % (N_WORDS, WIDTH)]
    for j in range(0, N_WORDS, 4):
        out.append(" ".join('"%s"' % (w + "\\0" * (WIDTH - len(w))) for w in words[j:j + 4]) + "\n")
    open(sys.argv[1], "w").write("".join(out))
    n_long = sum(len(w) > 16 for w in words)
    print(len(words), "words; longer than 16 bytes:", n_long, "; longest:", max(words, key=len), "; mean length %.2f" % (sum(map(len, words)) / len(words)))


if __name__ == "__main__":
    main()
