"""Would config 2 gain from going through K1p as two halves on two streams (each kernel's part-filled last round of workgroups
filled by the other half's kernels)?  Measured with the library as it is: 512 slices as one batch, as two batches of 256 one after
the other on one stream, and as the same two batches on two streams at once.  Run on the GPU box: python tools/two_streams.py"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
avr = importlib.import_module("avrecode-ms_amd")
import torch

dev = torch.device("cuda:0")
full = avr.DeviceWorkload.synth(2, 512, avr.KIND_CABAC, 0, 1000, 0)
PARTS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
halves = [avr.DeviceWorkload.synth(2, 512 // PARTS, avr.KIND_CABAC, 0, 1000, f * (512 // PARTS)) for f in range(PARTS)]
streams = [torch.cuda.Stream(dev) for _ in range(PARTS)]


def settle_all(ws):
    torch.cuda.synchronize(dev)
    for w in ws:
        assert not w.settle()["redone"]


def timed(fn, steps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3


def one():
    full.encode_chunked()


def in_turn():
    for w in halves:
        w.encode_chunked()


def at_once():
    for w, s in zip(halves, streams):
        with torch.cuda.stream(s):
            w.encode_chunked()


for w in [full] + halves:                      # the first run asks the device for the context count; the rest are sized by it
    w.encode_chunked()
settle_all([full] + halves)
at_once()
settle_all(halves)
print("one batch of 512 slices          %.3f ms" % timed(one))
print("%d parts, one after the other  %%.3f ms" % PARTS % timed(in_turn))
print("%d parts, on as many streams   %%.3f ms" % PARTS % timed(at_once))
settle_all([full] + halves)
a = [bytes(x) for x in full.results()[0]]
b = [bytes(x) for w in halves for x in w.results()[0]]
print("same bytes:", a == b)
