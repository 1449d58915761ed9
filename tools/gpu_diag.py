#!/usr/bin/env python
"""Run every GPU self-check, never stopping at the first failure; prints one line per check.
Usage on the GPU box:  python tools/gpu_diag.py > gpurun_out/diag.log 2>&1"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "kb-ner_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def run(name, fn):
    try:
        r = fn()
        print("[ok ] %-28s %s" % (name, r), flush=True)
        return r
    except Exception:
        print("[ERR] %-28s\n%s" % (name, traceback.format_exc()), flush=True)
        return None


def main():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    import selftest as st
    from kbner.lib import EPI_ADD, EPI_ATOMIC32, EPI_BIAS, EPI_DGELU, EPI_GELU, GEMM_NN, GEMM_NT, GEMM_TN
    print(torch.cuda.get_device_name(0), flush=True)

    def tr():
        got, exp = st.probe_tr()
        ok = bool(np.array_equal(got, exp))
        if not ok:
            print("tr-read got (lane: 4 values):")
            for lane in range(64):
                print(lane, got[lane].tolist(), "expected", exp[lane].tolist())
        return ok

    def mf():
        c, ref = st.probe_mfma()
        err = float((c - ref).abs().max())
        if err > 1e-2:
            print("mfma c:\n", c, "\nref:\n", ref, "\nref^T match:", float((c - ref.t()).abs().max()))
        return err

    run("probe_tr", tr)
    run("probe_mfma", mf)
    run("gemm NT 128", lambda: st.check_gemm(GEMM_NT, 128, 128, 64))
    run("gemm NT 256x384x192", lambda: st.check_gemm(GEMM_NT, 256, 384, 192))
    run("gemm NT bias+add", lambda: st.check_gemm(GEMM_NT, 256, 256, 128, EPI_BIAS | EPI_ADD))
    run("gemm NT bias+gelu", lambda: st.check_gemm(GEMM_NT, 256, 256, 128, EPI_BIAS | EPI_GELU))
    run("gemm NN 128", lambda: st.check_gemm(GEMM_NN, 128, 128, 64))
    run("gemm NN 256x384x192 dgelu", lambda: st.check_gemm(GEMM_NN, 256, 384, 192, EPI_DGELU))
    run("gemm NN add", lambda: st.check_gemm(GEMM_NN, 384, 128, 256, EPI_ADD))
    run("gemm TN 128", lambda: st.check_gemm(GEMM_TN, 128, 128, 64, EPI_ATOMIC32))
    run("gemm TN 256x384x512 sk4", lambda: st.check_gemm(GEMM_TN, 256, 384, 512, EPI_ATOMIC32, splitk=4))
    run("gemm NT 2048x1024x1024", lambda: st.check_gemm(GEMM_NT, 2048, 1024, 1024))
    run("attention B1 S64 A1", lambda: st.check_attention(1, 64, 1, ragged=False))
    run("attention B2 S128 A2", lambda: st.check_attention(2, 128, 2))
    run("attention B2 S512 A2", lambda: st.check_attention(2, 512, 2))
    run("layernorm 256x128", lambda: st.check_layernorm(256, 128))
    run("layernorm 300x768", lambda: st.check_layernorm(300, 768))
    run("layernorm 512x1024", lambda: st.check_layernorm(512, 1024))
    run("crf B3 n7", lambda: st.check_crf(3, 7))
    run("crf B32 n64", lambda: st.check_crf(32, 64))
    run("crf B8 n512", lambda: st.check_crf(8, 512))
    run("adamw", st.check_adamw)
    run("step tiny", st.check_step)


if __name__ == "__main__":
    main()
