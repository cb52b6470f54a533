"""A C-only host drives the GPU through include/kirag_amd.h: tests/capi/capi_gpu.c is compiled with gcc,
started as a process that contains neither Python nor torch (so /opt/rocm's HIP runtime serves it, not the torch wheel's), and runs
kr_index_create -> add -> search / search_async x 2 / finish_ex and kr_encoder_create_ex -> load_weight -> finalize -> forward on a tiny configuration.
The expected results in the vectors file come from the oracle (oracle/search_np.py: canonical top-k, bit-exact; oracle/encoder_np.py: e5_encode)."""
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def write_vectors(path):
    from oracle import encoder_np as E
    from oracle import search_np as S
    S.build()
    rng = np.random.default_rng(17)
    n, d, nq, k = 6000, 192, 37, 25
    x = rng.standard_normal((n, d)).astype(np.float32); x /= np.linalg.norm(x, axis=1, keepdims=True)
    q = x[rng.choice(n, nq)] + 0.1 * rng.standard_normal((nq, d)).astype(np.float32)
    x[4001] = x[13]                                              # an exact tie across the two add() pieces
    es, er = S.search_canonical(q, x, k)
    H, L, heads, FF, vocab, max_pos = 128, 2, 2, 512, 1000, 512
    w = E.synth_weights(H, L, FF, vocab, max_pos, seed=5)
    ids, mask = E.synth_tokens(9, 24, seed=9, ragged=True, vocab_lo=5, vocab_hi=vocab, min_len=3)
    emb = E.e5_encode(w, ids, mask, heads)
    with open(path, "wb") as f:
        f.write(b"KRT1" + struct.pack("<4i", n, d, nq, k))
        f.write(x.tobytes()); f.write(np.ascontiguousarray(q, np.float32).tobytes()); f.write(es.astype(np.float32).tobytes()); f.write(er.astype(np.int64).tobytes())
        f.write(struct.pack("<7if", H, L, heads, FF, vocab, max_pos, 2, 1e-12))
        f.write(struct.pack("<i", len(w)))
        for name, t in w.items():
            b = name.encode()
            f.write(struct.pack("<i", len(b)) + b + struct.pack("<q", t.size) + np.ascontiguousarray(t, np.float32).tobytes())
        f.write(struct.pack("<3i", ids.shape[0], ids.shape[1], 0))
        f.write(ids.astype(np.int64).tobytes()); f.write(mask.astype(np.int64).tobytes()); f.write(emb.astype(np.float32).tobytes())
        f.write(struct.pack("<f", 4e-3))                          # tiny configurations, element-wise bar of the default mode


def build_exe(tmp_path):
    exe = str(tmp_path / "capi_gpu")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(REPO, "include"),
                           os.path.join(REPO, "tests", "capi", "capi_gpu.c"), "-o", exe, "-ldl", "-lm"])
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_gpu_host_compiles_as_plain_c(tmp_path):
    build_exe(tmp_path)                                          # CPU suite: the program is C99 against the header alone


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_only_host_drives_index_and_encoder_on_the_gpu(tmp_path):
    exe = build_exe(tmp_path)
    vec = str(tmp_path / "vectors.bin")
    write_vectors(vec)
    env = {k: v for k, v in os.environ.items() if not k.startswith("KIRAG_AMD_")}
    out = subprocess.run([exe, os.path.join(REPO, "kirag_amd", "libkirag_amd.so"), vec], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "capi_gpu ok" in out.stdout
    assert out.stdout.count("forward_packed: bit-identical to forward") == 2      # both encoder modes, from the ragged token list (ABI 9)
