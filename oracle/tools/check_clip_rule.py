#!/usr/bin/env python3
"""Dev-time check (needs /root/reference, so it cannot run on the GPU box).

Parses the auto-generated switch of clip_polygon() in the reference
(src/shaders/polygon_clipping.glsl:35-225), evaluates every case symbolically
and compares the resulting vertex list with the *rule* that oracle/ and the HIP
kernel implement (see clip_rule() below and DESIGN.md "Clipping").  The rule is
our own restatement; this script only proves it reproduces the reference's
output order for every sign mask.
"""
import re, sys

REF = "/root/reference/src/shaders/polygon_clipping.glsl"


def parse_reference():
    cases = {}
    for line in open(REF):
        m = re.match(r"\s*case\s+(\d+):\s+vc = (\d+);(.*)break;", line)
        if not m:
            continue
        mask, vc, body = int(m.group(1)), int(m.group(2)), m.group(3)
        n = mask & 7
        v = ["v%d" % i for i in range(10)]
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            a = re.match(r"v\[(\d+)\] = v\[(\d+)\]$", stmt)
            b = re.match(r"v\[(\d+)\] = iz0\(v\[(\d+)\], v\[(\d+)\]\)$", stmt)
            if a:
                v[int(a.group(1))] = v[int(a.group(2))]
            elif b:
                l, r = v[int(b.group(2))], v[int(b.group(3))]
                assert l.startswith("v") and r.startswith("v"), (mask, stmt)
                v[int(b.group(1))] = "I(%s,%s)" % (l[1:], r[1:])
            else:
                raise SystemExit("unparsed: " + stmt)
        out = v[:vc]
        closure = v[vc] if vc > 0 else None
        cases.setdefault(mask, []).append((vc, out, closure))
    return cases


def clip_rule(n, above):
    """Our rule.  n = vertex count (3..7), above = list of bools (z > 0).

    Walk the polygon once starting at vertex 0, emit surviving vertices and, at
    every sign change, the crossing I(i, i+1) computed with the operands in
    polygon order.  Among the cyclic rotations of that sequence pick the one
    that leaves the most surviving vertices at their original array index (it
    needs the fewest register moves); ties go to the smallest rotation.
    Returns (count, list) or (0, []) when nothing survives or the above-set is
    not one contiguous run (non-convex input)."""
    k = sum(above)
    if k == 0:
        return 0, []
    if k == n:
        return n, ["v%d" % i for i in range(n)]
    changes = sum(above[i] != above[(i + 1) % n] for i in range(n))
    if changes != 2:
        return 0, []
    seq = []
    for i in range(n):
        if above[i]:
            seq.append("v%d" % i)
        if above[i] != above[(i + 1) % n]:
            seq.append("I(%d,%d)" % (i, (i + 1) % n))
    vc = len(seq)
    best = None
    for r in range(vc):
        out = [seq[(j + r) % vc] for j in range(vc)]
        writes = sum(1 for j in range(vc) if out[j] != "v%d" % j)
        # closure write v[vc] = v[0] unless that slot already holds out[0]
        writes += 0 if out[0] == "v%d" % vc else 1
        if best is None or writes < best[0]:
            best = (writes, r, out)
    return vc, best[2]


def main():
    cases = parse_reference()
    bad = 0
    total = 0
    for n in range(3, 8):
        for bits in range(1 << n):
            above = [(bits >> i) & 1 == 1 for i in range(n)]
            mask = n | (bits << 3)
            vc, out = clip_rule(n, above)
            ref = cases.get(mask)
            total += 1
            if ref is None:
                if vc != 0:
                    print("rule gives", vc, "but reference has no case", mask)
                    bad += 1
                continue
            for (rvc, rout, rclosure) in ref:
                if rvc != vc or rout != out:
                    print("MISMATCH mask", mask, "n", n, "ref", rvc, rout, "rule", vc, out)
                    bad += 1
                # when the reference writes/keeps a closure it must equal out[0]
                if vc and rclosure is not None and rclosure != out[0] and rclosure != "v%d" % vc:
                    print("closure mismatch", mask, rclosure, out[0])
                    bad += 1
    print("checked %d masks, %d reference cases, %d mismatches" % (total, sum(len(v) for v in cases.values()), bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
