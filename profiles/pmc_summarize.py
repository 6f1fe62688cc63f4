"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch per kernel family.

Corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of
the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B), so it is doubled; WRITE_SIZE is
exact for 16-B-per-lane stores. hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def family(name):
    if "gemv1_kernel" in name or "gemv_kernel" in name:
        return "gemv_kernel"  # the single-clip register kernel and the LDS kernel are one family (decode linears, <= 4 clips)
    if "decode_cgemm_kernel" in name or "decode_logits_kernel" in name:
        return "decode_gemm_kernel"  # batched decode linear layers: clip-block GEMMs + vocabulary projection (5+ clips)
    if "bf16_kernel" in name and "gemm" in name:
        return "gemm_bf16_kernel"  # encoder GEMM, all three tile shapes
    for f in ("decode_persistent_kernel", "decode_gemm_kernel", "decode_attention_kernel", "act_prep_kernel", "advance_kernel",
              "encoder_attention_kernel", "layernorm_bf16_kernel", "stft_mel_kernel", "mel_normalize_kernel"):
        if f in name:
            return f
    return None


def collect(d, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            fam = family(r["Kernel_Name"])
            if fam:
                tot[fam] += float(r["Counter_Value"])
                cnt[fam] += 1
    return tot, cnt


def main():
    out_dir, key = sys.argv[1], sys.argv[2]
    fetch, nf = collect(os.path.join(out_dir, "fetch"), "FETCH_SIZE")
    write, nw = collect(os.path.join(out_dir, "write"), "WRITE_SIZE")
    res = {}
    for fam in sorted(set(fetch) | set(write)):
        f_kib = fetch[fam] / max(nf[fam], 1)
        w_kib = write[fam] / max(nw[fam], 1)
        res[fam] = {"launches_sampled": nf[fam], "fetch_size_kib_raw_per_launch": round(f_kib, 2),
                    "write_size_kib_per_launch": round(w_kib, 2),
                    "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[3] if len(sys.argv) > 3 else "r02_pmc_traffic.json")
    allres = json.load(open(path)) if os.path.exists(path) else {}
    allres[key] = res
    json.dump(allres, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
