"""The C-ABI library loads, exports every symbol the header declares, and refuses to code
without a GPU (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "avrecode_ms_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avr_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(avr):
    names = declared_symbols()
    assert len(names) >= 25
    handle = ctypes.CDLL(avr.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/avrecode_ms_amd.h but not exported"
    assert set(names) == set(avr.SIGNATURES), set(names) ^ set(avr.SIGNATURES)


def test_tables_match_oracle(avr, oracle):
    assert avr.cabac_tables() == oracle.tables()


def test_host_epilogue_helpers_match_oracle(avr, oracle):
    for raw in (b"", b"\x80", b"\x12\x80", b"\x12\x34", b"\x80\x80", bytes(range(7))):
        assert avr.drop_stop_byte(raw) == oracle.drop_stop_byte(raw)
        for parity in (-1, 0, 1):
            assert avr.tail_patch(raw, parity, 0x5A) == oracle.tail_patch(raw, parity, 0x5A)


def test_no_cpu_fallback(avr):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    assert avr.device_count() == 0
    with pytest.raises(avr.AvrError, match="no HIP device"):
        avr.Batch(0, 4, 1024)
