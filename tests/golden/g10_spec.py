"""Inputs of golden set G10 (tests/golden/g10_encoder_large_ext.npz): full e5-large / bge-large shape beyond G2.  Only the reference's OUTPUTS are
stored; weights and token ids are regenerated from these tables on both sides (generator: make_golden.py g10, which imports the reference; consumer:
tests/test_gpu_encoder.py).  No reference code is involved in this module."""
import numpy as np

from oracle.encoder_np import synth_tokens

CFG = dict(H=1024, L=24, heads=16, FF=4096, vocab=30522, max_pos=512)

WEIGHTS = {          # name -> (recipe in oracle.encoder_np, kwargs)
    "benign": ("synth_weights", dict(seed=0)),
    "out3": ("synth_weights_outlier", dict(seed=7, gamma_lo=1.5, gamma_hi=3.0)),        # |x| up to ~80 against a median of 0.36: the "two orders of magnitude"
                                                                                        # of real BERT-family checkpoints' outlier channels
    "out16": ("synth_weights_outlier", dict(seed=7, gamma_lo=8.0, gamma_hi=16.0)),      # residual stream: |x| up to ~400 against a median of 0.34
    "out60": ("synth_weights_outlier", dict(seed=7, gamma_lo=30.0, gamma_hi=60.0)),     # |x| up to ~1600
}
CASES = {            # (B, S, layout, token seed): layout R = ragged right-padded, L = ragged left-padded, F = full length
    "benign": {"e5": [(2, 512, "R", 11), (4, 256, "L", 12), (64, 128, "R", 13)], "bge": [(2, 512, "R", 14), (4, 256, "L", 15)]},
    "out3": {"e5": [(8, 128, "R", 41), (16, 32, "R", 42), (2, 512, "R", 43), (4, 256, "L", 44)], "bge": [(8, 128, "R", 45), (4, 256, "L", 46)]},
    "out16": {"e5": [(8, 128, "R", 21), (2, 512, "R", 22), (4, 256, "L", 23), (16, 32, "R", 24)], "bge": [(8, 128, "R", 25), (4, 256, "L", 26)]},
    "out60": {"e5": [(8, 128, "R", 31), (16, 32, "R", 32), (2, 512, "R", 33)], "bge": [(8, 128, "R", 34)]},
}


def tokens(B, S, layout, seed):
    ids, mask = synth_tokens(B, S, seed=seed, ragged=layout != "F")
    if layout == "L":                       # the left-padding branch of the reference's truncate_to_max_sequence (dataset/collators.py:38-44)
        ids = np.ascontiguousarray(ids[:, ::-1]); mask = np.ascontiguousarray(mask[:, ::-1])
    return ids, mask


def weights(name):
    import oracle.encoder_np as enp
    recipe, kw = WEIGHTS[name]
    return getattr(enp, recipe)(CFG["H"], CFG["L"], CFG["FF"], CFG["vocab"], CFG["max_pos"], **kw)
