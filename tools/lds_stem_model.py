#!/usr/bin/env python
"""LDS bank-conflict model of the f16mx stem (csrc/conv.hip, vgg_stem_x3_kernel<true>) after the banking rules of
/opt/skills/guides/MI355X_MICROARCH.md §LDS: lane groups and bank modulus per instruction, one LDS cycle per group,
N distinct addresses on one bank within a group = N cycles.  Counts the EXTRA cycles per 8 x 32 tile and workgroup
(what SQ_LDS_BANK_CONFLICT counts) for the consumers' fragment reads and the producers' line writes, for the layout
of rounds 3-5 and for candidate fixes (round 6).   python tools/lds_stem_model.py
"""
import itertools

HW = 34
G128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G32x2 = [list(range(32)), list(range(32, 64))]
GW8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def cycles(groups, addrs, nbytes, mod):
    """(ideal, actual) LDS-array cycles of one wave instruction: addrs[lane] = byte address or None (inactive)."""
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addrs[l]
            if a is None:
                continue
            for b in range(a // 4, (a + nbytes) // 4):
                banks.setdefault(b % mod, set()).add(a)
        tot += max((len(s) for s in banks.values()), default=0)
    ideal = sum(1 for g in groups if any(addrs[l] is not None for l in g))
    return ideal, tot


def swz_old(hy, hx):
    return ((hx >> 1) & 7) ^ ((hy & 1) << 2)


def consumer_reads(swz, tail_b128):
    """extra cycles per tile: 4 waves x 2 passes x 9 taps of pixel + weight fragment reads"""
    extra = ideal_t = 0
    for wave in range(4):
        for ky, kx in itertools.product(range(3), range(3)):
            # pixel fragments, two blocks
            for i in range(2):
                def addr(lane, part, off=0):
                    half, l31 = lane >> 5, lane & 31
                    ly = 2 * wave + ((l31 >> 1) & 1)
                    lx = 2 * (l31 >> 2) + (l31 & 1) + 16 * i
                    hy, hx = ly + ky, lx + kx
                    slot = ({0: 0, 1: 2, 2: 5, 3: 7}[part] ^ half) ^ swz(hy, hx)
                    return (hy * HW + hx) * 128 + slot * 16 + off
                for part in (0, 1, 2):
                    a, b = cycles(G128, [addr(l, part) for l in range(64)], 16, 64)
                    ideal_t += a; extra += b - a
                if tail_b128:
                    a, b = cycles(G128, [addr(l, 3) for l in range(64)], 16, 64)
                    ideal_t += a; extra += b - a
                else:
                    a, b = cycles(G32x2, [addr(l, 3) for l in range(64)], 8, 64)
                    ideal_t += a; extra += b - a
                    a, b = cycles(G32x2, [addr(l, 3, 12) for l in range(64)], 4, 32)
                    ideal_t += a; extra += b - a
            # weight fragment (rows m = l31 of this tap's 32)
            def waddr(lane, kk, off=0):
                half, l31 = lane >> 5, lane & 31
                return l31 * 128 + (((2 * kk + half) ^ ((l31 >> 1) & 7)) << 4) + off
            for kk in (0, 1, 2):
                a, b = cycles(G128, [waddr(l, kk) for l in range(64)], 16, 64)
                ideal_t += a; extra += b - a
            if tail_b128:
                a, b = cycles(G128, [waddr(l, 3) for l in range(64)], 16, 64)
                ideal_t += a; extra += b - a
            else:
                a, b = cycles(G32x2, [waddr(l, 3) for l in range(64)], 8, 64)
                ideal_t += a; extra += b - a
                a, b = cycles(G32x2, [waddr(l, 3, 12) for l in range(64)], 4, 32)
                ideal_t += a; extra += b - a
    return 2 * ideal_t, 2 * extra          # two passes


def producer_writes(swz, perm):
    """extra LDS-array cycles per tile: 11 blocks x 2 channel halves of line writes"""
    extra = ideal_t = 0
    for pw in range(11):
        def row(l31):
            return 32 * pw + perm(l31)
        def addr(lane, slot, off=0, only_half=None):
            half, l31 = lane >> 5, lane & 31
            r = row(l31)
            if r >= 340 or (only_half is not None and half != only_half):
                return None
            hy, hx = divmod(r, HW)
            return r * 128 + ((slot(half) ^ swz(hy, hx)) << 4) + off
        ops = [(GW8, lambda h: 2 * h, 0, None, 16), (GW8, lambda h: 2 * h + 1, 0, None, 16),
               (GW8, lambda h: 4, 0, 0, 12), (GW8, lambda h: 5, 0, 0, 12),
               (G32x2, lambda h: 4, 12, 1, 4), (G32x2, lambda h: 5, 12, 1, 4),
               (GW8, lambda h: 6, 0, 1, 16), (GW8, lambda h: 7, 0, 1, 16)]
        for groups, slot, off, only, nb in ops:
            a, b = cycles(groups, [addr(l, slot, off, only) for l in range(64)], nb, 32)
            ideal_t += a; extra += b - a
    return 2 * ideal_t, 2 * extra


if __name__ == "__main__":
    ident = lambda l: l
    perm2 = lambda l: 2 * (l & 7) + ((l >> 3) & 1) + 16 * (l >> 4)
    for name, swz, tail, perm in (("rounds 3-5: b64 + b32 tails, lane = pixel", swz_old, False, ident),
                                  ("b128 tails", swz_old, True, ident),
                                  ("b128 tails, producer lanes 0-7 <-> even pixels", swz_old, True, perm2)):
        ri, re = consumer_reads(swz, tail)
        wi, we = producer_writes(swz, perm)
        print(f"{name:55s} reads ideal {ri:5d} + conflicts {re:5d} | writes ideal {wi:5d} + conflicts {we:5d} | "
              f"conflict cycles per tile {re + we}")
