"""The residual stream's low byte (kirag_amd/csrc/encoder.hip: lo_encode / lo_decode) as a numpy specification: x is kept as hi = the 16-bit operand
(round to nearest) plus ONE byte = the distance bits(x) - bits(hi) on the fp32 bit patterns in units of 2^LO_SH fp32 ulps (= ulp(hi) / 256), biased by 128
and clamped BEFORE the shift (the shift-then-clamp form is selected to v_ashr_pk_u8_i32, which writes only 16 bits on gfx950: tests/test_build_quality.py).
The properties below are what the kernels rely on; the device functions are exercised by tools/lo_codec_check.hip and, end to end, by the encoder goldens."""
import numpy as np
import pytest


def to_hi(x, kind):
    if kind == "f16":
        return x.astype(np.float16).astype(np.float32)
    b = x.view(np.uint32).astype(np.uint64)
    b = (b + 0x7fff + ((b >> 16) & 1)) >> 16 << 16          # bf16: round to nearest even on the bit pattern
    return b.astype(np.uint32).view(np.float32)


def encode(x, hi, sh):
    d = (x.view(np.uint32).astype(np.int64) - hi.view(np.uint32).astype(np.int64) + 2**31) % 2**32 - 2**31      # int32 wrap
    return (np.clip(d + (1 << (sh - 1)) + (128 << sh), 0, (256 << sh) - 1) >> sh).astype(np.uint8)


def decode(byte, hi, sh):
    """lo_decode_final: what the readers of the final hidden state use (a non-finite hi decodes to itself, round 5); for finite hi identical to the
    3-instruction lo_decode of the LayerNorm's residual path"""
    x = ((hi.view(np.uint32).astype(np.int64) + (byte.astype(np.int64) << sh) - (128 << sh)) % 2**32).astype(np.uint32).view(np.float32)
    return np.where((hi.view(np.uint32) & 0x7f800000) == 0x7f800000, hi, x)


@pytest.mark.parametrize("kind,sh,mant", [("f16", 5, 10), ("bf16", 8, 7)])
def test_low_byte_is_a_256th_of_the_operand_ulp(kind, sh, mant):
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(400_000) * np.exp(rng.uniform(-6, 3, 400_000))).astype(np.float32)
    x = np.concatenate([x, -x[:1000], np.float32([1.0, 2.0, 1.9999999, -1.9999999, 0.5, 3.0e-5, -3.0e-5])])
    hi = to_hi(x, kind)
    byte = encode(x, hi, sh)
    dec = decode(byte, hi, sh)
    with np.errstate(divide="ignore"):                         # hi == 0 for |x| < 2^-25: excluded by `normal` below
        ulp = np.exp2(np.floor(np.log2(np.abs(hi).astype(np.float64))) - mant)
    err_hi, err = np.abs(hi.astype(np.float64) - x), np.abs(dec.astype(np.float64) - x)
    normal = np.abs(x) > (6.2e-5 if kind == "f16" else 1e-30)
    assert (err <= err_hi + 1e-300).all()                         # never worse than the 16-bit operand alone
    assert (err[normal] <= ulp[normal] / 256 * 0.5001 + ulp[normal] / 256 * (byte[normal] == 255)).all()   # half a unit; one unit at the clamped tie
    assert np.mean(err[normal] / ulp[normal]) < 0.0015           # 8 more significand bits on average
    assert (np.sign(dec) == np.sign(x))[normal].all()
    # byte 128 is "no low half" (what the kernels read when the stream is kept without one, and what the workspace is initialised to)
    assert np.array_equal(decode(np.full(hi.shape, 128, np.uint8), hi, sh), hi)
    # decode(encode) is a fixed point: re-encoding a decoded value against the same hi reproduces the byte (the statistics of the fused stream rely on it)
    assert np.array_equal(encode(dec, hi, sh), byte)


@pytest.mark.parametrize("kind,sh", [("f16", 5), ("bf16", 8)])
def test_non_finite_operands_survive_the_codec(kind, sh):
    """with a LayerNorm output beyond the operand range hi = +-inf, the encoder clamps the byte to 0 and the pure bit-pattern decode returned
    bits(inf) - 4096 = a FINITE 3.4e38: the residual path swallowed the overflow (it was still caught through the 16-bit stream and the pooled norm).
    Now the readers of the FINAL hidden state (pooling, kr_encoder_last_hidden: lo_decode_final) hand a non-finite hi through whatever the byte — behind
    the last LayerNorm there is no GEMM that would carry the overflow — while the LayerNorm's own residual decode keeps the plain form: inside the stack the
    16-bit stream itself feeds the next GEMM and every LayerNorm behind it sees NaN through y (tests/test_gpu_lifecycle.py covers both places)."""
    x = np.float32([np.inf, -np.inf, np.nan, 1.0e38 if kind == "bf16" else 7.0e4, -7.0e4])
    with np.errstate(over="ignore", invalid="ignore"):
        hi = to_hi(x, kind)
    for b in (0, 1, 128, 255):
        dec = decode(np.full(hi.shape, b, np.uint8), hi, sh)
        nonfinite = ~np.isfinite(hi)
        assert np.array_equal(dec[nonfinite].view(np.uint32), hi[nonfinite].view(np.uint32))
    byte = encode(x, hi, sh)
    dec = decode(byte, hi, sh)
    assert np.isinf(dec[0]) and np.isinf(dec[1]) and dec[1] < 0 and np.isnan(dec[2])
    if kind == "f16":
        assert np.isinf(dec[3]) and np.isinf(dec[4])             # 7e4 > 65504: hi overflowed, and so does the decoded value
