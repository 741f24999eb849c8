"""Crafted streams against the slice parser of row f4 (avrecode-ms_amd/csrc/host/avr_h264.h), under AddressSanitizer and
UBSan on the CPU build of the `recode` command line.

The reference leaves header validation to libavcodec; this build parses SPS / PPS / slice headers itself, so every
Exp-Golomb value that becomes an int, an index or a size has to be bounded here (bit_reader::ue_max).  Each case below
used to reach an out-of-bounds access or undefined behaviour: first_mb_in_slice >= 2^31 (negative after the cast: passed
the range check and indexed the macroblock array), a code of 32 leading zeros (1u << 32), a picture of 2^32 macroblocks a
side, a seq_parameter_set_id >= 2^31 in a PPS (negative index into the SPS table).  They must all end as header failures:
the slices stay literal, the file still round-trips.
"""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "avrecode-ms_amd", "csrc")
PKG = os.path.join(ROOT, "avrecode-ms_amd")


class Bits:
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        self.bits += [(v >> (n - 1 - i)) & 1 for i in range(n)]

    def ue(self, v):
        v += 1
        n = v.bit_length()
        self.bits += [0] * (n - 1)
        self.u(n, v)

    def se(self, v):
        self.ue(2 * v - 1 if v > 0 else -2 * v)

    def rbsp(self):
        bits = self.bits + [1]
        bits += [0] * (-len(bits) % 8)
        raw = bytearray(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))
        out, zeros = bytearray(), 0
        for b in raw:                                   # emulation prevention, 7.4.1
            if zeros >= 2 and b <= 3:
                out.append(3)
                zeros = 0
            out.append(b)
            zeros = zeros + 1 if b == 0 else 0
        return bytes(out)


def nal(kind, ref, payload):
    return b"\0\0\0\1" + bytes([(ref << 5) | kind]) + payload


def sps(width_minus1=19, height_minus1=14):
    b = Bits()
    b.u(8, 100); b.u(8, 0); b.u(8, 30); b.ue(0)                       # High profile, level 3, id 0
    b.ue(1); b.ue(0); b.ue(0); b.u(1, 0); b.u(1, 0)                   # 4:2:0, 8 bit, no scaling matrices
    b.ue(0); b.ue(0); b.ue(0); b.ue(1); b.u(1, 0)                     # frame_num / poc of 4 bits, one reference
    b.ue(width_minus1); b.ue(height_minus1)
    b.u(1, 1); b.u(1, 1); b.u(1, 0); b.u(1, 0)                        # frame_mbs_only, direct_8x8, no cropping, no VUI
    return nal(7, 3, b.rbsp())


def pps(sps_id=0):
    b = Bits()
    b.ue(0); b.ue(sps_id); b.u(1, 1); b.u(1, 0); b.ue(0)              # CABAC, one slice group
    b.ue(0); b.ue(0); b.u(1, 0); b.u(2, 0); b.se(0); b.se(0); b.se(0); b.u(1, 0); b.u(1, 0); b.u(1, 0)
    return nal(8, 3, b.rbsp())


def idr_slice(first_mb=0, leading_zeros=None):
    b = Bits()
    if leading_zeros is None:
        b.ue(first_mb)
    else:
        b.bits += [0] * leading_zeros + [1] * (1 + min(leading_zeros, 40))
    b.ue(7); b.ue(0); b.u(4, 0); b.ue(0); b.u(4, 0); b.u(1, 0); b.u(1, 0); b.se(0)
    return nal(5, 3, b.rbsp() + bytes(range(1, 200)))


CASES = {
    "first_mb_in_slice_of_2_to_the_31": sps() + pps() + idr_slice(0x7ffffffe) + idr_slice(0x80000000) + idr_slice(0xfffffffe),
    "exp_golomb_code_of_32_zeros": sps() + pps() + idr_slice(leading_zeros=32) + idr_slice(leading_zeros=31) + idr_slice(leading_zeros=200),
    "picture_of_2_to_the_32_macroblocks_a_side": sps(0xfffffffe, 0xfffffffe) + pps() + idr_slice(0) + sps(2047, 2047) + pps() + idr_slice(5),
    "negative_seq_parameter_set_id": sps() + pps(0x80000005) + idr_slice(0) + pps(0xfffffffe) + idr_slice(0),
    "first_mb_just_past_the_picture": sps() + pps() + idr_slice(300) + idr_slice(299),
}


@pytest.fixture(scope="module")
def recode_asan(tmp_path_factory):
    """The CLI's host side with ASan + UBSan (CPU build only: sanitizers do not run on the GPU box).  It links the product
    library for its C ABI but none of these inputs reaches a batch: nothing parses, so nothing is coded."""
    import avrecode_ms_amd as avr
    avr.build_native()
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    out = str(tmp_path_factory.mktemp("asan") / "recode_asan")
    cmd = [cxx, "-O1", "-g", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-I" + CSRC,
           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", out, os.path.join(CSRC, "host", "recode_main.cpp"),
           "-L" + PKG, "-lavrecode_hip", "-Wl,-rpath," + PKG]
    subprocess.run(cmd, check=True)
    return out


@pytest.mark.parametrize("case", sorted(CASES))
def test_crafted_headers_are_rejected_not_dereferenced(recode_asan, tmp_path, case):
    src = tmp_path / (case + ".h264")
    src.write_bytes(CASES[case])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
    probe = subprocess.run([recode_asan, "probe", str(src)], capture_output=True, text=True, env=env, timeout=120)
    assert probe.returncode == 0, probe.stderr[-2000:]
    res = json.loads(probe.stdout)
    # no slice of these parses; a slice whose header is in range (the last of "just past the picture") fails in its payload
    assert res["parse_to_the_end"] == 0 and res["header_failures"] >= 1 and res["header_failures"] + res["fail"] >= res["slices"], probe.stdout
    assert "AddressSanitizer" not in probe.stderr and "runtime error" not in probe.stderr, probe.stderr[-2000:]
    assert sum(res["literal_reasons"].values()) >= res["slices"] and any(k.startswith("header: ") for k in res["literal_reasons"]), probe.stdout
    # and the file still round-trips: what does not parse stays literal (no GPU needed: there is nothing to code)
    comp, back = tmp_path / "c.recode", tmp_path / "back.h264"
    for cmd in ([recode_asan, "compress", str(src), str(comp)], [recode_asan, "decompress", str(comp), str(back)]):
        run = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=120)
        assert run.returncode == 0 and "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-2000:]
    assert back.read_bytes() == CASES[case]


def test_directory_driver_on_files_with_nothing_to_code(recode_asan, tmp_path):
    """`recode test <dir>` (test.cpp:113-148) on a directory of the crafted streams, three copies each: the driver that takes the
    directory's files through both directions together (host side one file per thread, one GPU batch per direction for all of them)
    writes, file for file, what the reference's loop writes (AVR_TEST_SEQUENTIAL=1).  Nothing in these files parses, so nothing is
    coded and no GPU is needed: this is the threads, the windows (AVR_TEST_WINDOW=4: several of them) and the bookkeeping, under ASan +
    UBSan; the same comparison on real clips, with the batches, is tests/test_h264.py::test_recode_test_directory_batches_across_files."""
    a, b = tmp_path / "together", tmp_path / "one_by_one"
    for d in (a, b):
        d.mkdir()
        for k in range(3):
            for case, data in CASES.items():
                (d / f"{k}_{case}.h264").write_bytes(data)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", AVR_TEST_WINDOW="4")
    for d, extra in ((a, {}), (b, {"AVR_TEST_SEQUENTIAL": "1"})):
        run = subprocess.run([recode_asan, "test", str(d)], capture_output=True, text=True, env=dict(env, **extra), timeout=600)
        assert run.returncode == 0 and "failed on" not in run.stdout, run.stdout + run.stderr[-2000:]
        assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-2000:]
    names = sorted(p.name for p in a.iterdir() if p.is_file())
    assert len(names) == 3 * len(CASES)
    for name in names:
        assert (a / "output" / name).read_bytes() == (b / "output" / name).read_bytes() and (a / "output" / name).stat().st_size > 0, name
    rows = (a / "output" / "metrics.csv").read_text().strip().splitlines()
    assert rows[0] == (b / "output" / "metrics.csv").read_text().splitlines()[0] and len(rows) == 1 + len(names)
    assert (a / "output" / "log.txt").read_text().count("Compress-decompress roundtrip succeeded:") == len(names)
