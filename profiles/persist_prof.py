"""Summarise AX_WHISPER_PERSIST_PROF output: per-phase time of the persistent batch-1 decode launch.

    AX_WHISPER_PERSIST_PROF=gpurun_out/pp.txt python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python profiles/persist_prof.py gpurun_out/pp.txt
"""
import sys

import numpy as np

NAMES = ["P qkv.gather", "P qkv.ln", "P sa.gather", "P sa.wait", "P o.gather", "P cq.gather", "P cq.ln", "P ca.gather", "P ca.wait",
         "P co.gather", "P co.merge", "P fc1.gather", "P fc1.ln", "P fc2.gather", "P lg.gather+ln", "P amax.gather",
         "C qkv.wait", "C qkv.rows", "C sa.all", "C o.wait", "C o.rows", "C cq.wait", "C cq.rows", "C ca.all", "C co.wait", "C co.rows",
         "C fc1.wait", "C fc1.rows", "C fc2.wait", "C fc2.prefetch", "C logits", "C fc2.rows"]


def main(path):
    head = open(path).readline().split()
    steps = int(head[2])
    a = np.loadtxt(path, comments="#")[:, : len(NAMES)] * 0.01 / steps  # us per decoder step
    print(f"steps {steps}, workgroups {a.shape[0]}; microseconds per decoder step (12 layers summed)")
    print(f"{'phase':16s} {'wg0':>8s} {'median':>8s} {'min':>8s} {'max':>8s}")
    for i, n in enumerate(NAMES):
        col = a[:, i]
        nz = col[col > 0] if (col > 0).any() else col
        print(f"{n:16s} {a[0, i]:8.2f} {np.median(nz):8.2f} {nz.min():8.2f} {nz.max():8.2f}")
    print(f"{'total P':16s} {a[0, :16].sum():8.2f} {np.median(a[:, :16].sum(1)):8.2f}")
    print(f"{'total C':16s} {a[0, 16:].sum():8.2f} {np.median(a[:, 16:].sum(1)):8.2f}")


if __name__ == "__main__":
    main(sys.argv[1])


# (query fold, round 5: "P cq gathered" = a cross-attention unit has its query (T + statistics gathered), "C cq published" = a row
# producer has published y1 / T / statistics; "P cq ln done" and "C cq rows start" do not exist there)
TL = ["P qkv gathered", "P qkv ln done", "P o gathered", "P cq gathered", "P cq ln done", "P co gathered", "P co merged",
      "P fc1 gathered", "P fc1 ln done", "P fc2 gathered",
      "C qkv published", "C sa published", "C o published", "C cq published", "C ca published", "C co published",
      "C fc1 published", "C fc2 published",
      "C qkv rows start", "C o rows start", "C cq rows start", "C co rows start", "C fc1 rows start", "C fc2 rows start"]


def timeline(path):
    """Absolute timeline of one layer in the middle of the utterance: min / median / max over workgroups, microseconds
    relative to the first event."""
    a = np.loadtxt(path, comments="#")
    if a.shape[1] < 64:
        return
    t = a[:, 32:32 + len(TL)] * 0.01
    t[t == 0] = np.nan
    base = np.nanmin(t)
    order = np.argsort(np.nanmedian(t, axis=0))
    print("\ntimeline of one layer (us after its first event): min / median / max over workgroups, n = workgroups that logged it")
    for i in order:
        col = t[:, i] - base
        if np.all(np.isnan(col)):
            continue
        print(f"{TL[i]:20s} {np.nanmin(col):7.2f} {np.nanmedian(col):7.2f} {np.nanmax(col):7.2f}   n={int(np.sum(~np.isnan(col)))}")


if __name__ == "__main__":
    timeline(sys.argv[1])
