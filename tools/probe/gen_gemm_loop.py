#!/usr/bin/env python3
"""Generates the K-loop body of the fp32 MFMA GEMM from an instruction schedule.

A schedule maps an MFMA slot (group 0..3 x index 0..15) to the memory operations issued right after that
MFMA.  Operations: G0..G7 global loads of the next tile (A rows 0..3, W rows 0..3), Rk.j = fragment read j
(0,1 = A, 2,3 = B) of k-block k (k = 4 means k-block 0 of the NEXT tile, legal only after the barrier),
W0..W7 = LDS writes of the next tile.  Even k-blocks live in the f registers, odd ones in g.
Prints one `#define LOOP_BODY_<name>` per schedule to stdout (used by gemm_sched.hip).
"""
import sys

SCHEDULES = {}


def sched(name, table, barrier=(2, 15), pf=1):
    SCHEDULES[name] = (table, barrier, pf)


def spread(ops, slots):
    return dict(zip(slots, [[o] for o in ops]))


def merge(*ds):
    out = {}
    for d in ds:
        for k, v in d.items():
            out.setdefault(k, []).extend(v)
    return out


G = [f"G{j}" for j in range(8)]
W = [f"W{j}" for j in range(8)]


def R(k):
    return [f"R{k}.{j}" for j in range(4)]


# A: the first interleaved variant (one op per MFMA, clustered at the start of each group)
sched("A", {0: merge(spread(G, range(0, 8)), spread(R(1), range(8, 12))), 1: spread(R(2), range(0, 4)),
            2: merge(spread(R(3), range(0, 4)), spread(W, range(4, 12))), 3: spread(R(4), range(0, 4))})
# B: one op every other MFMA where the dependences allow it
sched("B", {0: merge(spread(G, range(0, 16, 2)), spread(R(1), range(1, 8, 2))), 1: spread(R(2), range(0, 8, 2)),
            2: merge(spread(R(3), range(1, 8, 2)), spread(W, [0, 2, 4, 6, 8, 10, 12, 13])), 3: spread(R(4), range(0, 8, 2))})
# C: global loads spread over groups 0 and 1, everything else every other MFMA
sched("C", {0: merge(spread(G[:4], range(0, 8, 2)), spread(R(1), range(1, 8, 2))),
            1: merge(spread(R(2), range(0, 8, 2)), spread(G[4:], range(1, 8, 2))),
            2: merge(spread(R(3), range(1, 8, 2)), spread(W, [0, 2, 4, 6, 8, 10, 12, 13])), 3: spread(R(4), range(0, 8, 2))})
# D: like A but the LDS writes every other MFMA
sched("D", {0: merge(spread(G, range(0, 8)), spread(R(1), range(8, 12))), 1: spread(R(2), range(0, 4)),
            2: merge(spread(R(3), [1, 3, 5, 7]), spread(W, [0, 2, 4, 6, 8, 10, 12, 13])), 3: spread(R(4), range(0, 4))})
# E: reads late in the previous group (just in time), loads first
sched("E", {0: merge(spread(G, range(0, 8)), spread(R(1), range(9, 13))), 1: spread(R(2), range(8, 12)),
            2: merge(spread(W, range(0, 8)), spread(R(3), range(8, 12))), 3: spread(R(4), range(8, 12))})

# F: LDS writes as late as the dependences allow (barrier in the middle of k-block 3): more time for the global loads
sched("F", {0: merge(spread(G, range(0, 8)), spread(R(1), range(8, 12))), 1: spread(R(2), range(0, 4)),
            2: merge(spread(R(3), range(0, 4)), spread(W[:4], range(12, 16))),
            3: merge(spread(W[4:], range(0, 4)), spread(R(4), range(7, 11)))}, barrier=(3, 6))
# G: schedule A with prefetch distance 2 (two staging register sets, loop unrolled by 2)
sched("G", {0: merge(spread(G, range(0, 8)), spread(R(1), range(8, 12))), 1: spread(R(2), range(0, 4)),
            2: merge(spread(R(3), range(0, 4)), spread(W, range(4, 12))), 3: spread(R(4), range(0, 4))}, pf=2)
# H: F with prefetch distance 2
sched("H", {0: merge(spread(G, range(0, 8)), spread(R(1), range(8, 12))), 1: spread(R(2), range(0, 4)),
            2: merge(spread(R(3), range(0, 4)), spread(W[:4], range(12, 16))),
            3: merge(spread(W[4:], range(0, 4)), spread(R(4), range(7, 11)))}, barrier=(3, 6), pf=2)

# S: schedule H + the write-out of the PREVIOUS output tile (persistent kernel): per K-tile one 8-row pass of the C tile:
#    E0 = residual loads, E1 = C-staging read from LDS, E2 = bias/residual/ReLU + global store
sched("S", {0: merge(spread(G, range(0, 8)), spread(R(1), range(8, 12)), {13: ["E0"]}), 1: merge(spread(R(2), range(0, 4)), {8: ["E1"]}),
            2: merge(spread(R(3), range(0, 4)), {8: ["E2"]}, spread(W[:4], range(12, 16))),
            3: merge(spread(W[4:], range(0, 4)), spread(R(4), range(7, 11)))}, barrier=(3, 6), pf=2)

# M: LDS-DMA staging (global_load_lds_dwordx4, no staging registers, no ds_write): 8 DMA instructions per wave and K-tile
#    issued singly under k-block 0, `s_waitcnt vmcnt(0)` (VW) right before the barrier that publishes the tile
sched("M", {0: merge(spread([f"D{j}" for j in range(8)], range(0, 8)), spread(R(1), range(8, 12))), 1: spread(R(2), range(0, 4)),
            2: spread(R(3), range(0, 4)), 3: merge({5: ["VW"]}, spread(R(4), range(7, 11)))}, barrier=(3, 6), pf=1)

REGS = [["ra0", "ra1", "ra2", "ra3", "rb0", "rb1", "rb2", "rb3"], ["sa0", "sa1", "sa2", "sa3", "sb0", "sb1", "sb2", "sb3"]]


def emit_op(op, cur, lset, sset):
    if op[0] == "E":
        return f"EP{op[1]}();"
    if op[0] == "D":
        return f"DMA({op[1]}, {cur ^ 1});"
    if op == "VW":
        return "VMWAIT();"
    if op[0] == "G":
        j = int(op[1])
        base, ld = ("Ag", "lda") if j < 4 else ("Wg", "ldw")
        return f"GL({REGS[lset][j]}, {base}, {ld}, {j % 4}, knext);"
    if op[0] == "W":
        j = int(op[1])
        return f"SW({REGS[sset][j]}, {'As' if j < 4 else 'Bs'}, {cur ^ 1}, {j % 4});"
    k, j = int(op[1]), int(op[3])
    s = "f" if k % 2 == 0 else "g"
    reg = f"{s}{'a' if j < 2 else 'b'}{j % 2}"
    buf, kb = (str(cur ^ 1), 0) if k == 4 else (str(cur), k)
    base, off = ("As", "a_off") if j < 2 else ("Bs", "b_off")
    return f"FR({reg}, {base}, {off}, {buf}, {kb}, {j % 2});"


def emit(name):
    """Two bodies per schedule: _0 for even K-tiles (LDS buffer 0 is current), _1 for odd ones."""
    return emit_half(name, 0) + "\n" + emit_half(name, 1)


def emit_half(name, cur):
    table, barrier, pf = SCHEDULES[name]
    lset = cur if pf == 2 else 0          # pf 2: tile kt+2 is loaded into set (kt & 1), tile kt+1 stored from the other
    sset = 1 - cur if pf == 2 else 0
    lines = []
    for g in range(4):
        s = "f" if g % 2 == 0 else "g"
        for i in range(16):
            c = "xyzw"[i // 4]
            acc = ["acc00", "acc01", "acc10", "acc11"][i % 4]
            a = f"{s}a{(i % 4) // 2}.{c}"
            b = f"{s}b{(i % 4) % 2}.{c}"
            line = f"MM({acc}, {a}, {b}); SB;"
            for op in table.get(g, {}).get(i, []):
                line += " " + emit_op(op, cur, lset, sset) + " SB;"
            if (g, i) == barrier:
                line += " __syncthreads();"
            lines.append(line)
    return f"#define LOOP_BODY_{name}_{cur} \\\n" + " \\\n".join("  " + l for l in lines) + "\n"


# ---- 64 x 128 block tile (4 waves as 2 x 2, each 32 x 64 = 1 x 2 MFMA tiles) -------------------------------------------
# Used for launches that would give too few 128 x 128 tiles to fill the chip.  Per wave and K-tile: 32 MFMAs (4 k-blocks x 8),
# 6 global loads (A rows lrow, lrow+32; W rows lrow + 32 j), 12 fragment reads (A, B0, B1 per k-block), 6 LDS writes.
# Four accumulators (two output tiles x even/odd k-pairs) so that an accumulator is reused every 4th MFMA.
def emit64(cur):
    regs = [["ra0", "ra1", "rb0", "rb1", "rb2", "rb3"], ["sa0", "sa1", "sb0", "sb1", "sb2", "sb3"]]
    lset, sset = cur, 1 - cur          # prefetch distance 2, as schedule H

    def G(j):
        base, ld, jj = ("Ag", "lda", j) if j < 2 else ("Wg", "ldw", j - 2)
        return f"GL({regs[lset][j]}, {base}, {ld}, {jj}, knext);"

    def W(j):
        base, jj = ("As", j) if j < 2 else ("Bs", j - 2)
        return f"SW{'A' if j < 2 else 'B'}({regs[sset][j]}, {base}, {cur ^ 1}, {jj});"

    def R(k, j):                       # j: 0 = A, 1 = B0, 2 = B1
        s = "f" if k % 2 == 0 else "g"
        buf, kb = (cur ^ 1, 0) if k == 4 else (cur, k)
        if j == 0:
            return f"FRA({s}a, {buf}, {kb});"
        return f"FRB({s}b{j - 1}, {buf}, {kb}, {j - 1});"

    table = {0: {0: [R(1, 0)], 1: [R(1, 1)], 2: [R(1, 2)], 3: [G(0)], 4: [G(1)], 5: [G(2)], 6: [G(3)], 7: [G(4)]},
             1: {0: [R(2, 0)], 1: [R(2, 1)], 2: [R(2, 2)], 3: [G(5)]},
             2: {0: [R(3, 0)], 1: [R(3, 1)], 2: [R(3, 2)], 4: [W(0)], 5: [W(1)], 6: [W(2)], 7: [W(3)]},
             3: {0: [W(4)], 1: [W(5)], 3: [R(4, 0)], 4: [R(4, 1)], 5: [R(4, 2)]}}
    barrier = (3, 2)
    lines = []
    for g in range(4):
        s = "f" if g % 2 == 0 else "g"
        for i in range(8):
            c = "xyzw"[i // 2]
            par = "e" if (i // 2) % 2 == 0 else "o"
            t = i % 2
            line = f"MM(acc{t}{par}, {s}a.{c}, {s}b{t}.{c}); SB;"
            for op in table.get(g, {}).get(i, []):
                line += " " + op + " SB;"
            if (g, i) == barrier:
                line += " __syncthreads();"
            lines.append(line)
    return f"#define LOOP_BODY_T64_{cur} \\\n" + " \\\n".join("  " + l for l in lines) + "\n"


if __name__ == "__main__":
    names = [n for n in sys.argv[1:] if n != "T64"] or (sorted(SCHEDULES) if len(sys.argv) == 1 else [])
    print("// generated by gen_gemm_loop.py — do not edit")
    for n in names:
        print(emit(n))
    for n in names:
        print(f"#define LOOP_PF_{n} {SCHEDULES[n][2]}")
    if "T64" in sys.argv[1:] or len(sys.argv) == 1:
        print(emit64(0))
        print(emit64(1))
