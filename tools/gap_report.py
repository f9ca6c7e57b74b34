#!/usr/bin/env python3
"""Where the GPU idles inside a PPO epoch: from a rocprofv3 --kernel-trace database (rocpd sqlite), the epochs of the headline job
(25 launches of the policy chain kernel - 24 rollout steps + the bootstrap value - followed by 40 forward launches of the update) are
located in the dispatch stream and,
per epoch: span, busy time (union of the kernel intervals), and the idle gaps grouped by the pair (kernel before, kernel after).

    python tools/gap_report.py OUT/NAME_results.db [out.md] [--horizon 25] [--steps 40] | --window 0.6 1.0
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
    ap.add_argument("--horizon", type=int, default=25, help="chain-kernel launches per epoch")
    ap.add_argument("--steps", type=int, default=40, help="optimizer steps per epoch")
    ap.add_argument("--chain", default="mlp_chain_fwd_kernel")
    ap.add_argument("--forward", default="split_gemm_kernel<true, 5, 0, 4, true, 18, true>")
    ap.add_argument("--window", type=float, nargs=2, default=None, metavar=("FROM", "TO"),
                    help="no epoch detection: analyse the dispatches whose start lies in this fraction of the trace's time span "
                         "(e.g. 0.6 1.0 = the steady state at the end of a run)")
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    ks = sorted(cur.execute("select name, start, end from kernels"), key=lambda r: r[1])
    if a.window:
        t0, t1 = ks[0][1], ks[-1][1]
        lo, hi = t0 + a.window[0] * (t1 - t0), t0 + a.window[1] * (t1 - t0)
        seg = [k for k in ks if lo <= k[1] <= hi]
        busy, cur_end, prev = 0.0, seg[0][1], None
        pair_tot, pair_cnt, kern_tot, kern_cnt = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
        for n, s0, e in seg:
            if prev is not None and s0 > cur_end:
                pair_tot[(short(prev), short(n))] += (s0 - cur_end) / 1e3
                pair_cnt[(short(prev), short(n))] += 1
            busy += (max(e, cur_end) - max(s0, cur_end)) / 1e3
            kern_tot[short(n)] += (e - s0) / 1e3
            kern_cnt[short(n)] += 1
            if e > cur_end:
                cur_end, prev = e, n
        span = (cur_end - seg[0][1]) / 1e3
        out = [f"# GPU idle time, window {a.window[0]:.2f} - {a.window[1]:.2f} of the trace ({len(seg)} dispatches)\n",
               f"span {span:.0f} us, busy {busy:.0f} us, idle {span - busy:.0f} us ({100 * (span - busy) / span:.1f} %)\n",
               "| before | after | gaps | idle us | us / gap |\n|---|---|---|---|---|"]
        for (p_, n_), t in pair_tot.most_common(25):
            out.append(f"| `{p_}` | `{n_}` | {pair_cnt[(p_, n_)]} | {t:.0f} | {t / pair_cnt[(p_, n_)]:.1f} |")
        out.append("\n| kernel | launches | us | us / launch |\n|---|---|---|---|")
        for n_, t in kern_tot.most_common(25):
            out.append(f"| `{n_}` | {kern_cnt[n_]} | {t:.0f} | {t / kern_cnt[n_]:.2f} |")
        # one example neighbourhood of each of the three largest gap classes: the 8 dispatches before and after the gap
        for (p_, n_), _ in pair_tot.most_common(3):
            cur_end, prev, hit = seg[0][1], None, None
            for i, (n, s0, e) in enumerate(seg):
                if prev is not None and s0 > cur_end and (short(prev), short(n)) == (p_, n_) and i > len(seg) // 2:
                    hit = (i, (s0 - cur_end) / 1e3)
                    break
                if e > cur_end:
                    cur_end, prev = e, n
            if hit:
                i, g = hit
                out.append(f"\nexample: `{p_}` -> `{n_}` ({g:.0f} us idle before dispatch {i}); dispatches {i - 8} .. {i + 7}, (start - gap end) us | duration us | kernel:\n")
                for n, s0, e in seg[i - 8:i + 8]:
                    out.append(f"    {(s0 - seg[i][1]) / 1e3:10.1f} | {(e - s0) / 1e3:8.1f} | {short(n)}")
        text = "\n".join(out) + "\n"
        if a.out:
            open(a.out, "w").write(text)
        print(text)
        return
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
