"""avrecode-ms_amd: MI355X-native arithmetic re-encoder for H.264 CABAC bin streams.

Python is plumbing here: it builds and loads the C-ABI shared library
(``libavrecode_hip.so``, declared in ``include/avrecode_ms_amd.h``) and offers thin
helpers that hand torch device memory to it.  All coding happens in the HIP kernels
(``csrc/avr_kernels.hip``); there is no CPU coding path in this package -- if the
library is missing or no GPU is visible the calls raise.

Reference surface mirrored (pbluc/avrecode-ms, /root/reference):
  * ``Batch``  -- deferred form of ``cabac::encoder`` (cabac_code.h:26-82, driven from
    recode.cpp:1442-1481) and ``recoded_code::encoder`` (recode.cpp:1270, 1075-1103).
  * ``drop_stop_byte`` / ``tail_patch`` -- recode.cpp:1508-1512 and :1354-1360.

The directory name contains a hyphen (it follows the reference's repository name);
``import avrecode_ms_amd`` works through the alias module at the repository root.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import shutil
import subprocess
import sys
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libavrecode_hip.so")
HOOKS_LIB_PATH = os.path.join(_HERE, "libavrecode_hip_hooks.so")   # -DAVR_TEST_HOOKS build, for tests/ only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "avrecode_ms_amd.h")

KIND_CABAC, KIND_RANGE, KIND_CABAC_CODES, KIND_CABAC8 = 0, 1, 2, 3
SEL_BYPASS, SEL_TERMINATE = 1024, 1025
SEL8_BYPASS, SEL8_TERMINATE, MAX_STATES8 = 126, 127, 126      # one-byte records (KIND_CABAC8)
SLICE_OK, SLICE_ZERO_PROB, SLICE_OVERFLOW, SLICE_BAD_RECORD = 0, 1, 2, 3
NOP_CABAC, NOP_RANGE = 1026 << 1, 0
CHUNK_BINS, SORT_BLOCK_BINS = 1024, 4096

_SOURCES = ["avr_kernels.hip", "avr_k1p.hip", "avr_k2p.hip", "avr_api.cpp"]
_DEPS = _SOURCES + ["avr_coder.h", "avr_div.h", "avr_internal.h", "avr_k1p.h", "avr_k2p.h", "avr_synth.h", "avr_tables.h"]


class AvrError(RuntimeError):
    """Raised for every negative return code of the C ABI (message = avr_last_error())."""


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP kernels and the C ABI for gfx950 (in-tree): libavrecode_hip.so, the product, and
    libavrecode_hip_hooks.so, the same sources with -DAVR_TEST_HOOKS (the switches tests/ use to force rare paths;
    the product library has no such switch).  One hipcc per source file, in parallel."""
    deps = [os.path.join(_CSRC, d) for d in _DEPS] + [HEADER_PATH]
    if not force and os.path.exists(LIB_PATH) and os.path.exists(HOOKS_LIB_PATH):
        if all(min(os.path.getmtime(LIB_PATH), os.path.getmtime(HOOKS_LIB_PATH)) >= os.path.getmtime(d) for d in deps):
            return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB_PATH):      # GPU box without a compiler in PATH: use the shipped build
            return LIB_PATH
        raise AvrError("hipcc not found and no prebuilt libavrecode_hip.so")
    objdir = os.path.join(_HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall"]
    jobs = []
    for variant, extra in (("", []), (".hooks", ["-DAVR_TEST_HOOKS"])):
        for src in _SOURCES:
            obj = os.path.join(objdir, src + variant + ".o")
            jobs.append((obj, [hipcc] + flags + extra + ["-c", "-o", obj, os.path.join(_CSRC, src)]))
    if verbose:
        for _, cmd in jobs:
            print(" ".join(cmd))
    procs = [subprocess.Popen(cmd) for _, cmd in jobs]
    if any(p.wait() != 0 for p in procs):
        raise AvrError("hipcc failed")
    for variant, path in (("", LIB_PATH), (".hooks", HOOKS_LIB_PATH)):
        objs = [os.path.join(objdir, src + variant + ".o") for src in _SOURCES]
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", path + ".tmp"] + objs, check=True)
        os.replace(path + ".tmp", path)
    return LIB_PATH


RECODE_PATH = os.path.join(_HERE, "recode")
_HOST = os.path.join(_CSRC, "host")


def build_recode(force: bool = False, verbose: bool = False) -> str:
    """Compile the `recode` command line (csrc/host/recode_main.cpp: the reference's CLI, recode.cpp:1642-1677, on this
    build's H.264 syntax parser and GPU batches) next to libavrecode_hip.so, which it links."""
    deps = [os.path.join(_HOST, f) for f in ("recode_main.cpp", "avr_h264.h", "avr_h264_tables.h", "avr_recode.h", "avr_host.h", "avr_model.h")]
    deps += [HEADER_PATH, LIB_PATH]
    if not force and os.path.exists(RECODE_PATH) and all(os.path.getmtime(RECODE_PATH) >= os.path.getmtime(d) for d in deps):
        return RECODE_PATH
    cxx = shutil.which("g++") or shutil.which("hipcc")
    if not cxx:
        if os.path.exists(RECODE_PATH):
            return RECODE_PATH
        raise AvrError("no C++ compiler and no prebuilt recode binary")
    cmd = [cxx, "-O2", "-pthread", "-std=c++17", "-I" + _CSRC, "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", RECODE_PATH + ".tmp",
           os.path.join(_HOST, "recode_main.cpp"), "-L" + _HERE, "-lavrecode_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(RECODE_PATH + ".tmp", RECODE_PATH)
    return RECODE_PATH


class ChunkPlan(ctypes.Structure):
    """avr_chunk_plan: device arrays of the intra-slice parallel path (include/avrecode_ms_amd.h)."""
    _fields_ = [("res_off", c_void_p), ("chunk_base", c_void_p), ("chunk_slice", c_void_p), ("blk_base", c_void_p),
                ("blk_slice", c_void_p), ("dig_off", c_void_p), ("res_total", c_uint64), ("dig_total", c_uint64),
                ("total_chunks", c_uint32), ("total_blocks", c_uint32)]


class ChunkedPart(ctypes.Structure):
    """avr_chunked_part: one part of avr_cabac_encode_chunked_device_parts (include/avrecode_ms_amd.h)."""
    _fields_ = [("rec_off", c_void_p), ("n_bins", c_void_p), ("n_slices", c_size_t), ("init_states", c_void_p),
                ("plan", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
                ("out_off", c_void_p), ("out_len", c_void_p), ("status", c_void_p), ("final_states", c_void_p),
                ("rows_hint", c_uint32), ("counts", c_void_p)]


MAX_PARTS = 8


class SynthConfig(ctypes.Structure):
    _fields_ = [("workload", c_int), ("scale_permille", c_uint32), ("seed", c_uint64),
                ("first_slice", c_uint64), ("n_states", c_uint32)]


_lib = None

# name -> (restype, argtypes); every symbol include/avrecode_ms_amd.h declares
_u8p, _u16p, _u32p, _u64p, _i32p = (POINTER(c_uint8), POINTER(c_uint16), POINTER(c_uint32),
                                    POINTER(c_uint64), POINTER(c_int32))
SIGNATURES = {
    "avr_last_error": (c_char_p, []),
    "avr_version": (c_char_p, []),
    "avr_device_count": (c_int, []),
    "avr_cabac_lps_range_table": (_u8p, []),
    "avr_cabac_mlps_state_table": (_u8p, []),
    "avr_batch_create": (c_void_p, [c_int, c_size_t, c_size_t]),
    "avr_batch_destroy": (None, [c_void_p]),
    "avr_batch_reset": (c_int, [c_void_p]),
    "avr_batch_add_slice_cabac": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    "avr_batch_add_slice_range": (c_int, [c_void_p, c_void_p, c_size_t]),
    "avr_batch_add_slice_codes": (c_int, [c_void_p, c_void_p, c_size_t]),
    "avr_batch_add_slice_cabac8": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    "avr_batch_reserve_slice": (c_int, [c_void_p, c_int, c_size_t, c_void_p, c_size_t, POINTER(c_void_p)]),
    "avr_batch_submit": (c_int, [c_void_p]),
    "avr_batch_wait": (c_int, [c_void_p]),
    "avr_batch_run": (c_int, [c_void_p]),
    "avr_batch_get": (c_int, [c_void_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int)]),
    "avr_batch_get_states": (c_int, [c_void_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t)]),
    "avr_batch_timings": (c_int, [c_void_p, POINTER(c_float)]),
    "avr_batch_run_info": (c_int, [c_void_p, POINTER(ctypes.c_uint32)]),
    "avr_multi_create": (c_void_p, [c_void_p, c_size_t, c_size_t, c_size_t]),
    "avr_multi_destroy": (None, [c_void_p]),
    "avr_multi_add_slice_cabac": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    "avr_multi_add_slice_range": (c_int, [c_void_p, c_void_p, c_size_t]),
    "avr_multi_add_slice_codes": (c_int, [c_void_p, c_void_p, c_size_t]),
    "avr_multi_run": (c_int, [c_void_p]),
    "avr_multi_get": (c_int, [c_void_p, c_size_t, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_int)]),
    "avr_multi_placement": (c_int, [c_void_p, c_size_t]),
    "avr_multi_load": (c_int, [c_void_p, c_void_p]),
    "avr_pack_tiles_device": (c_int, [c_int, c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                      c_void_p, c_void_p, c_void_p]),
    "avr_cabac_encode_tiles_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                              c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "avr_range_encode_tiles_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "avr_cabac_chunked_workspace_bytes": (c_size_t, [c_size_t, c_size_t, c_void_p]),
    "avr_cabac_encode_chunked_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                                                c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                                c_void_p]),
    "avr_cabac_encode_chunked_device_hinted": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                                                       c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                                       c_void_p, c_uint32, c_void_p]),
    "avr_cabac_encode_chunked_second_pass_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                                                            c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                                            c_void_p]),
    "avr_cabac_encode_tiles_device_hinted": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                     c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_uint32, c_void_p]),
    "avr_cabac_encode_chunked_device_parts": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t]),
    "avr_range_chunked_workspace_bytes": (c_size_t, [c_size_t, c_void_p, ctypes.c_uint64]),
    "avr_range_encode_chunked_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t,
                                                c_void_p, c_void_p, ctypes.c_uint64, c_void_p, c_void_p]),
    "avr_cabac_resolve_workspace_bytes": (c_size_t, [c_size_t, c_size_t, c_void_p]),
    "avr_cabac_resolve_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t,
                                         c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "avr_cabac_resolved_workspace_bytes": (c_size_t, [c_size_t, c_void_p]),
    "avr_cabac_encode_resolved_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t,
                                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "avr_cabac_encode_codes_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "avr_cabac_encode_slices_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                               c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "avr_range_encode_slices_device": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "avr_context_census_device": (c_int, [c_int, c_void_p, c_void_p, c_uint64, c_void_p]),
    "avr_context_remap_device": (c_int, [c_int, c_void_p, c_void_p, c_uint64, c_void_p]),
    "avr_states_permute_device": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t,
                                          c_size_t, c_int]),
    "avr_synth_config_init": (c_int, [POINTER(SynthConfig), c_int, c_uint32, c_uint64]),
    "avr_synth_count_host": (c_int, [POINTER(SynthConfig), c_int, c_size_t, c_void_p]),
    "avr_synth_generate_host": (c_int, [POINTER(SynthConfig), c_int, c_size_t, c_void_p, c_void_p, c_void_p]),
    "avr_synth_count_device": (c_int, [c_int, c_void_p, POINTER(SynthConfig), c_int, c_size_t, c_void_p]),
    "avr_synth_generate_slices_device": (c_int, [c_int, c_void_p, POINTER(SynthConfig), c_int, c_size_t, c_void_p,
                                                 c_void_p, c_void_p]),
    "avr_synth_generate_tiles_device": (c_int, [c_int, c_void_p, POINTER(SynthConfig), c_int, c_size_t, c_void_p,
                                                c_void_p, c_void_p, c_void_p]),
    "avr_drop_stop_byte": (c_size_t, [c_void_p, c_size_t]),
    "avr_tail_patch": (c_size_t, [c_void_p, c_size_t, c_int, c_uint8]),
}


def _load(path):
    # One HIP runtime per process: torch bundles its own libamdhip64.so.7; loading it first
    # makes the dynamic linker bind this library to the same copy (same SONAME), which is
    # required anyway since device pointers and streams are shared with torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        handle = ctypes.CDLL(path)
    except OSError as exc:
        raise AvrError(f"cannot load the HIP extension {path}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)          # AttributeError = a declared symbol is missing
        fn.restype, fn.argtypes = res, args
    return handle


def lib():
    """Load libavrecode_hip.so (building it first if a compiler is present). Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_native()
        _lib = _load(LIB_PATH)
    return _lib


_hooks_lib = None


def library_sha256() -> str:
    """SHA-256 of the product library as it stands on disk (what profiles/pmc_traffic.json keys its counters by)."""
    import hashlib
    h = hashlib.sha256()
    with open(LIB_PATH, "rb") as f:
        for block in iter(lambda: f.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


@contextlib.contextmanager
def test_hooks(**hooks):
    """FOR tests/ ONLY.  Inside the block, lib() is libavrecode_hip_hooks.so -- the same sources built with
    -DAVR_TEST_HOOKS -- with the named hooks set (csrc/avr_internal.h: k1p_force_retry_every, census_stride,
    chain_lanes, k1_form_ref, k1_path [1 serial, 2 chunked], no_dense, no_hint, k2p_seg_len, local_waves, k2p_wave); all of them keep the bytes exact and
    only force paths that real batches take rarely.  The product library has no such switches."""
    global _lib, _hooks_lib
    if _hooks_lib is None:
        if not os.path.exists(HOOKS_LIB_PATH):
            build_native()
        _hooks_lib = _load(HOOKS_LIB_PATH)
        _hooks_lib.avr_test_hook_set.restype, _hooks_lib.avr_test_hook_set.argtypes = c_int, [c_char_p, c_uint32]
    saved, _lib = _lib, _hooks_lib
    try:
        _hooks_lib.avr_test_hook_set(b"reset", 0)
        for name, value in hooks.items():
            if _hooks_lib.avr_test_hook_set(name.encode(), int(value)) != 0:
                raise AvrError(_hooks_lib.avr_last_error().decode())
        yield _hooks_lib
    finally:
        _hooks_lib.avr_test_hook_set(b"reset", 0)
        _lib = saved


def _check(rc: int) -> int:
    if rc < 0:
        raise AvrError(f"avr error {rc}: {lib().avr_last_error().decode()}")
    return rc


def device_count() -> int:
    return lib().avr_device_count()


def cabac_tables():
    """(lps_range[512], mlps_state[256]) in the layout cabac_code.h:11-12 indexes."""
    L = lib()
    return (bytes(L.avr_cabac_lps_range_table()[:512]), bytes(L.avr_cabac_mlps_state_table()[:256]))


def drop_stop_byte(buf: bytes) -> bytes:
    """decompressor::cabac_decoder::finish (recode.cpp:1508-1512)."""
    a = (c_uint8 * max(len(buf), 1)).from_buffer_copy(buf.ljust(1, b"\0"))
    return bytes(buf[:lib().avr_drop_stop_byte(a, len(buf))])


def tail_patch(buf: bytes, length_parity: int, last_byte: int) -> bytes:
    """decompressor::run block patch (recode.cpp:1354-1360)."""
    a = (c_uint8 * (len(buf) + 1))()
    a[:len(buf)] = buf
    n = lib().avr_tail_patch(a, len(buf), length_parity, last_byte)
    return bytes(a[:n])


def make_cabac_records(bins, sels):
    """uint16 K1 records from bin values and selectors (numpy arrays or sequences)."""
    import numpy as np
    return (np.asarray(bins, dtype=np.uint16) & 1) | (np.asarray(sels, dtype=np.uint16) << 1)


def make_range_records(bins, pos, neg):
    """uint16 K2 records from bin values and the {pos,neg} estimator of each bin."""
    import numpy as np
    return ((np.asarray(bins, dtype=np.uint16) & 1) | (np.asarray(pos, dtype=np.uint16) << 1)
            | (np.asarray(neg, dtype=np.uint16) << 8))


class Batch:
    """Host-memory batch: add slices, run() on the GPU, get() the coded bytes.

    Mirrors how the reference's drivers use their coder objects: one encoder per slice
    (recode.cpp:1270, 1525), results read after all decoding (recode.cpp:1131, 1352-1363).
    """

    def __init__(self, device: int = 0, max_slices: int = 1024, max_bins: int = 1 << 24):
        self._L = lib()
        self._h = self._L.avr_batch_create(device, max_slices, max_bins)
        if not self._h:
            raise AvrError(self._L.avr_last_error().decode())

    def close(self):
        if self._h:
            self._L.avr_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may be gone already: its own teardown frees what is left
        if not sys.is_finalizing():
            self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def reset(self):
        _check(self._L.avr_batch_reset(self._h))

    def add_slice_cabac(self, recs, init_states) -> int:
        import numpy as np
        r = np.ascontiguousarray(recs, dtype=np.uint16)
        s = np.ascontiguousarray(init_states, dtype=np.uint8)
        return _check(self._L.avr_batch_add_slice_cabac(self._h, r.ctypes.data, r.size, s.ctypes.data, s.size))

    def add_slice_range(self, recs) -> int:
        import numpy as np
        r = np.ascontiguousarray(recs, dtype=np.uint16)
        return _check(self._L.avr_batch_add_slice_range(self._h, r.ctypes.data, r.size))

    def add_slice_cabac8(self, recs8, init_states) -> int:
        """K1 from one-byte records (bin | dense selector << 1; selectors SEL8_BYPASS / SEL8_TERMINATE): half the bytes over PCIe."""
        import numpy as np
        r = np.ascontiguousarray(recs8, dtype=np.uint8)
        s = np.ascontiguousarray(init_states, dtype=np.uint8)
        return _check(self._L.avr_batch_add_slice_cabac8(self._h, r.ctypes.data, r.size, s.ctypes.data, s.size))

    def add_codes(self, codes) -> int:
        """K1 from resolved codes (one uint8 per bin: AVR_CODE_CONTEXT / _BYPASS / _TERMINATE)."""
        import numpy as np
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        return _check(self._L.avr_batch_add_slice_codes(self._h, c.ctypes.data, c.size))

    def reserve(self, kind: int, n: int, init_states=None):
        """Zero-copy add: (slice index, numpy view of n elements in the batch's pinned staging buffer)."""
        import numpy as np
        st = None if init_states is None else np.ascontiguousarray(init_states, dtype=np.uint8)
        p = c_void_p()
        idx = _check(self._L.avr_batch_reserve_slice(self._h, kind, n, None if st is None else st.ctypes.data,
                                                     0 if st is None else st.size, ctypes.byref(p)))
        ctype = ctypes.c_uint8 if kind in (KIND_CABAC_CODES, KIND_CABAC8) else ctypes.c_uint16
        view = np.ctypeslib.as_array(ctypes.cast(p.value, POINTER(ctype)), shape=(n,)) if n else np.zeros(0, ctype)
        return idx, view

    def run_info(self):
        """{'chunked', 'rows_guessed', 'contexts_seen', 'ran_again'} of the last run."""
        v = (ctypes.c_uint32 * 4)()
        _check(self._L.avr_batch_run_info(self._h, v))
        return {"chunked": int(v[0]), "rows_guessed": int(v[1]), "contexts_seen": int(v[2]), "ran_again": int(v[3])}

    def submit(self):
        _check(self._L.avr_batch_submit(self._h))

    def wait(self):
        _check(self._L.avr_batch_wait(self._h))

    def run(self):
        _check(self._L.avr_batch_run(self._h))

    def get(self, i: int):
        """(bytes, status) of slice i."""
        p, n, st = c_void_p(), c_size_t(), c_int()
        _check(self._L.avr_batch_get(self._h, i, ctypes.byref(p), ctypes.byref(n), ctypes.byref(st)))
        return (ctypes.string_at(p.value, n.value) if n.value else b""), st.value

    def get_states(self, i: int) -> bytes:
        p, n = c_void_p(), c_size_t()
        _check(self._L.avr_batch_get_states(self._h, i, ctypes.byref(p), ctypes.byref(n)))
        return ctypes.string_at(p.value, n.value) if n.value else b""

    def timings(self):
        ms = (c_float * 4)()
        _check(self._L.avr_batch_timings(self._h, ms))
        return dict(zip(("h2d_ms", "pack_ms", "encode_ms", "d2h_ms"), list(ms)))


class MultiBatch:
    """One batch sharded over several GPUs (avr_multi_*): LPT by bin count, a host thread and an avr_batch per device."""

    def __init__(self, devices, max_slices: int = 1024, max_bins: int = 1 << 24):
        import numpy as np
        self._L = lib()
        self.devices = list(devices)
        d = np.asarray(self.devices, dtype=np.int32)
        self._h = self._L.avr_multi_create(d.ctypes.data, d.size, max_slices, max_bins)
        if not self._h:
            raise AvrError(self._L.avr_last_error().decode())

    def close(self):
        if self._h:
            self._L.avr_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        if not sys.is_finalizing():
            self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def add_slice_cabac(self, recs, init_states) -> int:
        import numpy as np
        r = np.ascontiguousarray(recs, dtype=np.uint16)
        s = np.ascontiguousarray(init_states, dtype=np.uint8)
        return _check(self._L.avr_multi_add_slice_cabac(self._h, r.ctypes.data, r.size, s.ctypes.data, s.size))

    def add_slice_range(self, recs) -> int:
        import numpy as np
        r = np.ascontiguousarray(recs, dtype=np.uint16)
        return _check(self._L.avr_multi_add_slice_range(self._h, r.ctypes.data, r.size))

    def add_codes(self, codes) -> int:
        import numpy as np
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        return _check(self._L.avr_multi_add_slice_codes(self._h, c.ctypes.data, c.size))

    def run(self):
        _check(self._L.avr_multi_run(self._h))

    def get(self, i: int):
        p, n, st = c_void_p(), c_size_t(), c_int()
        _check(self._L.avr_multi_get(self._h, i, ctypes.byref(p), ctypes.byref(n), ctypes.byref(st)))
        return (ctypes.string_at(p.value, n.value) if n.value else b""), st.value

    def placement(self, i: int) -> int:
        return _check(self._L.avr_multi_placement(self._h, i))

    def load(self):
        import numpy as np
        out = np.zeros(len(self.devices), dtype=np.uint64)
        _check(self._L.avr_multi_load(self._h, out.ctypes.data))
        return out.tolist()


from .device import DeviceWorkload, encode_tiles, plan_tiles, synth_config  # noqa: E402  (torch-backed helpers)

__all__ = ["AvrError", "Batch", "MultiBatch", "build_recode", "RECODE_PATH", "DeviceWorkload", "KIND_CABAC", "KIND_RANGE", "SEL_BYPASS", "SEL_TERMINATE",
           "build_native", "cabac_tables", "device_count", "drop_stop_byte", "encode_tiles", "lib",
           "make_cabac_records", "make_range_records", "plan_tiles", "synth_config", "tail_patch"]
