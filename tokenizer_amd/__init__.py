"""tokenizer_amd -- MI355X-native batch BPE encoder, drop-in for the Encode path of microsoft/Tokenizer.

    from tokenizer_amd import TokenizerBuilder
    tok = TokenizerBuilder.CreateTokenizer(open("cl100k_base.tiktoken", "rb").read(), specials, REGEX_CL100K)
    ids = tok.Encode("Hello World")                 # ITokenizer.Encode
    batch = tok.EncodeBatch(list_of_strings)        # the batch entry point

Everything below the class surface is libtkz (tokenizer_amd/csrc, C ABI in include/tkz.h): hand-written
HIP kernels for gfx950.  There is no CPU implementation in this package.
"""
from .tokenizer import (ENCODERS, MODEL_PREFIX_TO_ENCODING, MODEL_TO_ENCODING, REGEX_CL100K, REGEX_O200K, REGEX_PATTERN_1,
                        TikTokenizer, TokenizerBuilder, host_unicode_classes)

from .shardfile import Shard, write_shard

__all__ = ["Shard", "write_shard", "TikTokenizer", "TokenizerBuilder", "REGEX_PATTERN_1", "REGEX_CL100K", "REGEX_O200K", "ENCODERS",
           "MODEL_TO_ENCODING", "MODEL_PREFIX_TO_ENCODING", "host_unicode_classes"]
