#!/usr/bin/env python3
"""Fixture generator: the VALUES of the three literal arrays h264_model::get_model_key looks positions and block
categories up in (recode.cpp:691-704: sig_coeff_flag_offset_8x8[2][63], cat_lookup[14], sig_coeff_offset_dc[7]), read from
the reference's source text where it lies and written as data (tests/golden/model_tables.json).  avr_model.h re-derives
them (Table 9-43's frame column as inc_8x8_frame, Tables 9-34 / 9-40 as cat_base, Min(numDecod / 2, 2) for 4:2:2 chroma
DC); tests/test_host_model.py compares, and walks get_model_key over every position class against keys computed from
these values by the formula of recode.cpp:805-807 and :815.  Run in the build container (the reference is not on the GPU box)."""
import json
import os
import re

SRC = "/root/reference/recode.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_tables.json")


def ints(text):
    return [int(eval(e)) for e in text.split(",") if e.strip()]


def main():
    lines = open(SRC).read().split("\n")
    text = " ".join(l.split("//")[0] for l in lines[690:704])            # recode.cpp:691-704
    m8 = re.search(r"sig_coeff_flag_offset_8x8\[2\]\[63\]\s*=\s*\{\s*\{([^{}]*)\}\s*,\s*\{([^{}]*)\}\s*\}", text)
    mc = re.search(r"cat_lookup\[14\]\s*=\s*\{([^{}]*)\}", text)
    md = re.search(r"sig_coeff_offset_dc\[7\]\s*=\s*\{([^{}]*)\}", text)
    frame, field = ints(m8.group(1)), ints(m8.group(2))
    cat, dc = ints(mc.group(1)), ints(md.group(1))
    assert (len(frame), len(field), len(cat), len(dc)) == (63, 63, 14, 7)
    json.dump({"source": "recode.cpp:691-704 of pbluc/avrecode-ms (get_model_key, PIP_SIGNIFICANCE_MAP)",
               "sig_coeff_flag_offset_8x8": [frame, field], "cat_lookup": cat, "sig_coeff_offset_dc": dc},
              open(OUT, "w"), separators=(",", ":"))
    print("wrote", OUT)


if __name__ == "__main__":
    main()
