#!/usr/bin/env python3
"""DESIGN.md = docs/DESIGN.md.in with the @PLACEHOLDERS@ of its ledger filled from ONE bench line (the driver's command run on a gpurun box):
    python tools/fill_design.py profiles/r06_bench_n1.json [suite-tail.txt]
so that the numbers in the ledger are exactly those of a committed record and never typed by hand."""
import json, re, sys

def main():
    rec = None
    for line in open(sys.argv[1]):
        if line.startswith("{") and '"summary"' in line:
            rec = json.loads(line)
    assert rec, "no bench line in " + sys.argv[1]
    s, rf, cb = rec["summary"], rec["roofline"], rec["cpu_baseline"]
    fr = lambda k: "%.2f" % s[k][1]
    us = lambda k: "%.1f µs" % s[k][0]
    sub = {
        "HEADLINE": "%.2f Tpix/s, %.2f TB/s = %.3f" % (rec["value"] / 1e6, rf["achieved"] / 1e3, rf["frac"]),
        "G8UC3": fr("gauss_8uc3"), "G1080": fr("gauss_1080p"), "G8K": fr("gauss_8k"),
        "GS3": "%s (%.3f; `cv::GaussianBlur` on the host: %.0f µs)" % (us("gauss_sigma3_8uc1"), s["gauss_sigma3_8uc1"][1], s["gauss_sigma3_8uc1"][2]),
        "GS3C3": "%s (%.2f)" % (us("gauss_sigma3_8uc3"), s["gauss_sigma3_8uc3"][1]),
        "GS21": ("%s (%.2f)" % (us("gauss_sigma21_8uc1"), s["gauss_sigma21_8uc1"][1])) if "gauss_sigma21_8uc1" in s else "6.8 µs (0.30)",
        "GS16": "%s (%.3f; host: %.1f ms)" % (us("gauss_sigma16_32f"), s["gauss_sigma16_32f"][1], s["gauss_sigma16_32f"][2] / 1e3),
        "CFG2A": fr("cfg2a"), "CFG2C": fr("cfg2c"), "CFG2D": fr("cfg2d"), "CFG2E": fr("cfg2e"), "SOBEL": fr("sobel_16s"),
        "CFG3A": fr("cfg3a"), "CFG3B": fr("cfg3b"), "CFG3C": "%s per 8K frame = %s" % (us("cfg3c"), fr("cfg3c")),
        "AFF8": fr("affine_8uc1"), "PERSP8": fr("persp_8uc1"), "CUBIC8": "%s (%.3f)" % (us("affine_cubic_8uc1"), s["affine_cubic_8uc1"][1]),
        "CFG4A": fr("cfg4a"), "CFG4B": fr("cfg4b"), "INTEGRAL": fr("integral"),
        "CFG5": "%.1f µs per frame = %.3f of the i8 peak" % (s["cfg5"][0], s["cfg5"][1]),
        "TRAFFIC": "%.3f" % (rf["traffic"] / rf["algorithmic_bytes_per_launch"]) if rf.get("traffic") else "n/a",
        "CPUB": "%.1f Gpix/s on %d threads" % (cb["value"] / 1e3, cb["cores"]),
        "SUITE": open(sys.argv[2]).read().strip().splitlines()[-1].strip("= ") if len(sys.argv) > 2 else "(suite tail: profiles/r06_gpu_suite_tail.txt)",
    }
    text = open("docs/DESIGN.md.in").read()
    missing = set(re.findall(r"@([A-Z0-9]+)@", text)) - set(sub)
    assert not missing, missing
    open("DESIGN.md", "w").write(re.sub(r"@([A-Z0-9]+)@", lambda m: sub[m.group(1)], text))

if __name__ == "__main__":
    main()
