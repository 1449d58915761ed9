"""Measurement only (needs tools/experiments/gemm256_four_wave.patch applied and the library rebuilt): a few GEMM shapes through
libkbner_hip under the current KBNER_GEMM_W4 (bit mask of layouts) / KBNER_GEMM_W4_VAR environment -- the four-wave
128 x 128-per-wave structure of the 256 x 256 x 64 tile against the eight-wave kernel.  VAR 1 (no in-loop DMA) and VAR 2 (no
MFMA) are timing-only ablations with wrong results by design.  Output of round 3: profiles/round3_gemm_w4_lab.txt."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "kb-ner_amd"))
from kbner import ops  # noqa: E402
from kbner.lib import EPI_RMW32, GEMM_NN, GEMM_NT, GEMM_TN  # noqa: E402


def timed(fn, reps=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*s):
        return (torch.randn(*s, generator=g) * 0.05).to(torch.bfloat16).to(dev)

    out = []
    for M, N, K in ((8192, 8192, 8192), (65536, 1024, 4096), (65536, 4096, 1024)):
        X, W = rnd(M, K), rnd(N, K)
        C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        t = timed(lambda: ops.gemm(GEMM_NT, X, W, M, N, K, C=C, occupancy=True))
        out.append("NT %dx%dx%d %.0f" % (M, N, K, 2.0 * M * N * K / t / 1e12))
    M, N, K = 65536, 1024, 4096
    X, W = rnd(M, K), rnd(K, N)
    C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    t = timed(lambda: ops.gemm(GEMM_NN, X, W, M, N, K, C=C, occupancy=True))
    out.append("NN %dx%dx%d %.0f" % (M, N, K, 2.0 * M * N * K / t / 1e12))
    M, N, K = 65536, 4096, 1024
    dY, Xa = rnd(M, N), rnd(M, K)
    C32 = torch.zeros((N, K), dtype=torch.float32, device=dev)
    t = timed(lambda: ops.gemm_grouped(GEMM_TN, [ops.make_problem(dY, Xa, N, K, M, C32=C32, epi=EPI_RMW32)]))
    out.append("TN %dx%dx%d %.0f" % (N, K, M, 2.0 * M * N * K / t / 1e12))
    print("W4=%s VAR=%s | " % (os.environ.get("KBNER_GEMM_W4", "0"), os.environ.get("KBNER_GEMM_W4_VAR", "0")) + " | ".join(out))


if __name__ == "__main__":
    main()
