"""Probe 2: bf16 GEMMs on a stream masked to 224 CUs (7 of every 8), LSTM + small kernels on the default stream."""
import ctypes
import os
import sys
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
    a = torch.randn(8192, 3072, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(3072, 3072, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(1 << 16, device="cuda")

    def gemms(n=40):
        for _ in range(n):
            torch.matmul(a, w.t())

    def smalls(n=300):
        y = x
        for _ in range(n):
            y = y * 1.0001
        return y

    def timed(fn, stream):
        with torch.cuda.stream(stream):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); fn(); e1.record(stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    full = torch.cuda.Stream()
    print("gemms alone, unmasked: %.3f ms" % timed(gemms, full), flush=True)
    print("300 small kernels alone: %.3f ms" % timed(smalls, full), flush=True)
    for name, words in (("7of8", [0xfefefefe] * 8), ("hi224", [0, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff]), ("none", None)):
        s_big = masked_stream(words) if words else torch.cuda.Stream()
        print(f"{name}: gemms alone on it {timed(gemms, s_big):.3f} ms", flush=True)
        s2 = torch.cuda.Stream()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        with torch.cuda.stream(s_big):
            e[0].record(s_big); gemms(); e[1].record(s_big)
        with torch.cuda.stream(s2):
            e[2].record(s2); smalls(); e[3].record(s2)
        torch.cuda.synchronize()
        print(f"{name}: concurrent gemms {e[0].elapsed_time(e[1]):.3f} ms, 300 smalls {e[2].elapsed_time(e[3]):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
