#!/usr/bin/env python3
"""Heuristic lint for a hipcc (ROCm 7.2) code-generation hazard seen in hdk_part_aggregate.

A wave-uniform boolean that the compiler materialises as a lane mask with a VALU compare
(`v_cmp_*_e64 s[a:b], ...`) is only valid for the lanes that were active at the compare.  When that
compare sits inside a loop whose EXEC shrinks trip by trip and the mask later feeds a uniform branch
(`s_and[n2]_b64 vcc, exec, s[a:b]`) that runs under a different EXEC, lanes that had left the loop take
the wrong side.  (baseline_table.h: TableShape documents the instance that was hit.)

Usage: lint_lane_masks.py [file.s] [kernel-name-substring ...]   -- without a .s file it
cross-compiles the kernel sources to ISA (about
two minutes, no GPU needed) and lists, per kernel, every uniform branch whose mask comes from a VALU
compare, when the compare sits in the own lines of a loop that narrows EXEC (not merely around such a loop) and
the branch comes after that loop.
"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hdk_amd", "csrc")


def main():
    want = [a for a in sys.argv[1:] if not a.endswith(".s")]
    given = [a for a in sys.argv[1:] if a.endswith(".s")]
    if given:
        text = open(given[0]).read()
    else:
        text = ""
        with tempfile.TemporaryDirectory() as tmp:
            for src in ("scan_agg.hip", "scan_baseline.hip", "scan_project.hip", "reduce.hip", "join_build.hip", "init_groups.hip"):
                out = os.path.join(tmp, src + ".s")
                subprocess.check_call(["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-munsafe-fp-atomics",
                                       "-ffp-contract=off", "--cuda-device-only", "-S", src, "-o", out],
                                      cwd=CSRC, stderr=subprocess.DEVNULL)
                text += open(out).read()
    for m in re.finditer(r"^(_Z\w+|hdk_\w+):[^\n]*\n(.*?)\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want and not any(w in name for w in want):
            continue
        lines = body.splitlines()
        # loop extents from the compiler's own remarks ("in Loop: Header=BBn_m", "Parent Loop BBn_m");
        # a loop counts as divergent when EXEC is narrowed inside it
        labels = []  # (line, label, remark)
        for n, line in enumerate(lines, 1):
            lm = re.match(r"(\.LBB\d+_\d+):", line)
            if lm:
                k, remark = n, line
                while k < len(lines) and lines[k].lstrip().startswith(";"):
                    remark += lines[k]
                    k += 1
                labels.append((n, lm.group(1), remark))
        loops = []
        for idx, (n, lab, remark) in enumerate(labels):
            if "Loop Header" not in remark:
                continue
            tag = lab[2:]  # BBn_m
            inner = {tag} | {l2[2:] for (_, l2, r2) in labels
                             if "Loop Header" in r2 and re.search(r"Parent Loop " + tag + r"\b", r2)}
            member = [i2 for i2, (n2, lab2, r2) in enumerate(labels)
                      if lab2[2:] in inner or any(re.search(r"Header=" + t + r"\b", r2) for t in inner)]
            start = min(labels[i2][0] for i2 in member)
            last = max(member)
            end = labels[last + 1][0] - 1 if last + 1 < len(labels) else len(lines)
            loops.append((start, end, lab))
        # a loop narrows EXEC trip by trip when the narrowing sits in its OWN lines: a block-uniform loop around a
        # divergent inner loop runs its own lines under the EXEC it was entered with (the compare of a uniform value
        # at the head of a tile loop is complete, whatever the row loops below it do)
        narrowing = []
        for start, end, lab in loops:
            inner = [(s2, e2) for (s2, e2, l2) in loops if l2 != lab and start <= s2 and e2 <= end]
            own = [l for n2, l in enumerate(lines[start - 1:end], start) if not any(s2 <= n2 <= e2 for s2, e2 in inner)]
            if any("s_andn2_b64 exec, exec" in l for l in own):
                narrowing.append((start, end, lab))
        # innermost first: a compare belongs to the smallest loop around it
        loops = sorted(narrowing, key=lambda t: t[1] - t[0])
        last_def, hits = {}, []
        for n, line in enumerate(lines, 1):
            d = re.match(r"\s+(v_cmp\w+_e64)\s+(s\[\d+:\d+\])", line)
            if d:
                last_def[d.group(2)] = ("valu", n)
                continue
            d = re.match(r"\s+(s_\w+_b64)\s+(s\[\d+:\d+\])", line)
            if d and not d.group(1).startswith("s_cbranch"):
                last_def[d.group(2)] = ("salu", n)
            u = re.match(r"\s+s_andn?2?_b64 vcc, exec, (s\[\d+:\d+\])", line)
            if u:
                src = last_def.get(u.group(1))
                if src and src[0] == "valu":
                    for start, end, h in loops:
                        if start <= src[1] <= end and n > end:
                            hits.append((n, u.group(1), src[1], h, "after it"))
                            break
        if hits:
            print(name)
            for n, reg, dn, dl, ul in hits:
                print(f"  line {n}: uniform branch on {reg}  <- v_cmp at line {dn} (loop {dl}); use in loop {ul}")


if __name__ == "__main__":
    main()
