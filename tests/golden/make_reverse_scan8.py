#!/usr/bin/env python3
"""Fixture generator: the VALUES of the reference's scan_8 / reverse_scan_8 tables (recode.cpp:270-284, :286-319),
read from the reference's source text where it lies and written as data (tests/golden/reverse_scan8.json).
The build's neighbour geometry (avr_model.h: neighbor_block) is bit arithmetic; this pins it to the table the
reference actually looks neighbours up in (get_neighbor_sub_mb, recode.cpp:426-478).  Run in the build container
(the reference is not present on the GPU box; the JSON is what travels)."""
import json
import os
import re

SRC = "/root/reference/recode.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reverse_scan8.json")


def main():
    lines = open(SRC).read().split("\n")
    scan_txt = " ".join(lines[269:284])                      # constexpr uint8_t scan_8[...] = { ... };
    body = scan_txt[scan_txt.index("{") + 1:scan_txt.rindex("}")]
    scan_8 = [int(eval(e)) for e in body.split(",") if e.strip()]
    assert len(scan_8) == 16 * 3 + 3
    rev_txt = " ".join(l.split("//")[0] for l in lines[285:319])
    rev_txt = rev_txt[rev_txt.index("{") + 1:rev_txt.rindex("}")]
    cells = re.findall(r"r_scan8::inv\(\)|\{([^{}]*)\}", rev_txt)
    raw = re.findall(r"r_scan8::inv\(\)|\{[^{}]*\}", rev_txt)
    table = []
    for c in raw:
        if c.startswith("r_scan8"):
            table.append([0, True, True])                    # r_scan8::inv(), recode.cpp:247-249
        else:
            idx, left, up = [x.strip() for x in c.strip("{}").split(",")]
            table.append([int(eval(idx)), left == "true", up == "true"])
    assert len(table) == 15 * 8, len(table)
    rows = [table[8 * r:8 * r + 8] for r in range(15)]
    json.dump({"source": "recode.cpp:270-284 (scan_8), :286-319 (reverse_scan_8) of pbluc/avrecode-ms",
               "scan_8": scan_8, "reverse_scan_8": rows}, open(OUT, "w"), indent=0)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
