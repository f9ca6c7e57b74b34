#!/usr/bin/env python3
"""Where the GPU idles inside a PPO epoch: from a rocprofv3 --kernel-trace database (rocpd sqlite), the epochs of the headline job
(16 rollout launches of the policy chain kernel followed by 32 forward launches of the update) are located in the dispatch stream and,
per epoch: span, busy time (union of the kernel intervals), and the idle gaps grouped by the pair (kernel before, kernel after).

    python tools/gap_report.py OUT/NAME_results.db [out.md] [--horizon 16] [--steps 32]
"""
import argparse
import collections
import re
import sqlite3


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("out", nargs="?")
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--chain", default="mlp_chain_fwd_kernel")
    ap.add_argument("--forward", default="split_gemm_kernel<true, 5, 0, 4, true, 18, true>")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    ks = sorted(cur.execute("select name, start, end from kernels"), key=lambda r: r[1])
    marks = []          # (index into ks, 'C' | 'F')
    for i, (n, s, e) in enumerate(ks):
        if a.chain in n:
            marks.append((i, "C"))
        elif a.forward in n:
            marks.append((i, "F"))
    stream = "".join(m for _, m in marks)
    pat = re.compile("C{%d}F{%d}" % (a.horizon, a.steps))
    epochs = []
    for m in pat.finditer(stream):
        first = marks[m.start()][0]
        epochs.append(first)
    lines = []
    if len(epochs) < 3:
        raise SystemExit(f"only {len(epochs)} epochs found in the dispatch stream ({stream[:200]}...)")
    # epoch k spans from its first chain launch to the next epoch's first chain launch: use consecutive ones only
    per = []
    pair_tot = collections.Counter()
    pair_cnt = collections.Counter()
    kern_tot = collections.Counter()
    kern_cnt = collections.Counter()
    nep = 0
    for k in range(len(epochs) - 1):
        i0, i1 = epochs[k], epochs[k + 1]
        seg = ks[i0:i1]
        span = (ks[i1][1] - seg[0][1]) / 1e3
        if span > 60000:            # something else ran in between (kernel benches): not an epoch-to-epoch interval
            continue
        nep += 1
        busy = 0.0
        cur_end = seg[0][1]
        prev = None
        for n, s, e in seg + [ks[i1]]:
            if prev is not None and s > cur_end:
                g = (s - cur_end) / 1e3
                pair_tot[(short(prev), short(n))] += g
                pair_cnt[(short(prev), short(n))] += 1
            if (n, s, e) != ks[i1]:
                busy += (max(e, cur_end) - max(s, cur_end)) / 1e3
                kern_tot[short(n)] += (e - s) / 1e3
                kern_cnt[short(n)] += 1
                if e > cur_end:
                    cur_end = e
                    prev = n
        per.append((span, busy, len(seg)))
    lines.append(f"# GPU idle time inside the PPO epoch ({nep} consecutive epochs of the timed region)\n")
    lines.append("| epoch | span us | busy us | idle us | dispatches |\n|---|---|---|---|---|")
    for k, (sp, b, n) in enumerate(per):
        lines.append(f"| {k} | {sp:.0f} | {b:.0f} | {sp - b:.0f} | {n} |")
    sp = sum(p[0] for p in per) / nep
    b = sum(p[1] for p in per) / nep
    lines.append(f"\nmean: span {sp:.0f} us, busy {b:.0f} us, idle {sp - b:.0f} us ({100 * (sp - b) / sp:.1f} %)\n")
    lines.append("idle gaps by (kernel before -> kernel after), per epoch:\n")
    lines.append("| before | after | gaps / epoch | idle us / epoch | us / gap |\n|---|---|---|---|---|")
    for (p, n), t in pair_tot.most_common(30):
        c = pair_cnt[(p, n)]
        lines.append(f"| `{p}` | `{n}` | {c / nep:.1f} | {t / nep:.1f} | {t / c:.2f} |")
    lines.append("\nkernel time per epoch:\n")
    lines.append("| kernel | launches / epoch | us / epoch | us / launch |\n|---|---|---|---|")
    for n, t in kern_tot.most_common(30):
        lines.append(f"| `{n}` | {kern_cnt[n] / nep:.1f} | {t / nep:.1f} | {t / kern_cnt[n]:.2f} |")
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
