"""Feasibility probe: LSTM cluster kernels on a CU-masked stream beside bf16 GEMMs on the complementary mask."""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from aptai_amd import ops

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    torch.cuda.init()
    B, T, Tp = 16, 499, 512
    g = torch.Generator(device="cuda").manual_seed(0)
    xproj = torch.randn(B * Tp, 2048, device="cuda", generator=g) * 0.1
    whh = torch.randn(2, 1024, 256, device="cuda", generator=g) * 0.05
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    a = torch.randn(8192, 3072, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(3072, 3072, device="cuda", dtype=torch.bfloat16)

    def lstm():
        return ops.lstm_fwd(xproj, whh, lens, B, Tp, T)

    def gemms(n=40):
        for _ in range(n):
            torch.matmul(a, w.t())

    def timed(fn, stream):
        with torch.cuda.stream(stream):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); fn(); e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    full = torch.cuda.Stream()
    print("lstm alone, unmasked: %.3f ms" % timed(lstm, full), flush=True)
    print("gemms alone, unmasked: %.3f ms" % timed(gemms, full), flush=True)
    for name, small, big in () if os.environ.get("PROBE_MASKS", "0") != "1" else (("low32", [0xffffffff, 0, 0, 0, 0, 0, 0, 0], [0, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff]),
                             ("every8th", [0x01010101] * 8, [0xfefefefe] * 8),
                             ("low64", [0xffffffff, 0xffffffff, 0, 0, 0, 0, 0, 0], [0, 0, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff])):
        s_small, s_big = masked_stream(small), masked_stream(big)
        t_l = timed(lstm, s_small)
        t_g = timed(gemms, s_big)
        # concurrent
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        with torch.cuda.stream(s_big):
            e[0].record(s_big); gemms(); e[1].record(s_big)
        with torch.cuda.stream(s_small):
            e[2].record(s_small); h, _, _ = lstm(); e[3].record(s_small)
        torch.cuda.synchronize()
        st = ops.lstm_status(xproj.device)
        print(f"{name}: lstm on small mask {t_l:.3f} ms | gemms on big mask {t_g:.3f} ms | concurrent: gemms {e[0].elapsed_time(e[1]):.3f} lstm {e[2].elapsed_time(e[3]):.3f} "
              f"span {e[0].elapsed_time(e[3]):.3f} status {st} finite {bool(torch.isfinite(h).all())}", flush=True)
    # unmasked concurrency
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    with torch.cuda.stream(s1):
        e[0].record(s1); gemms(); e[1].record(s1)
    with torch.cuda.stream(s2):
        e[2].record(s2); h, _, _ = lstm(); e[3].record(s2)
    torch.cuda.synchronize()
    print(f"unmasked concurrent: gemms {e[0].elapsed_time(e[1]):.3f} lstm {e[2].elapsed_time(e[3]):.3f} finite {bool(torch.isfinite(h).all())}", flush=True)


if __name__ == "__main__":
    main()
