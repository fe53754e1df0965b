#!/usr/bin/env python3
"""CPU check of conv_wf4_kernel's LDS halo image (nd_conv_winograd_f4.hip): the unit a DMA lane fills must be the unit
every (tile, patch element, k group) reads, for both block geometries, and a ds_read_b64 must see at most 2-way bank conflicts."""
import itertools
def geo(GW):
    if GW == 5: return dict(TWL2=2, THL2=2, NIBL=0)
    return dict(TWL2=1, THL2=1, NIBL=2)
def run(GW):
    g = geo(GW); TWL2, THL2, NIBL = g['TWL2'], g['THL2'], g['NIBL']
    TW, TH = 4 << TWL2, 4 << THL2; HW, HH = TW + 2, TH + 2
    NG = (1 << NIBL) * HH * GW; NDMA = (NG + 47) // 48
    def key(li, hyq, gxq):
        if GW == 5: return ((hyq & 3) << 2) | (gxq & 3)
        return ((li & 3) << 2) | ((hyq & 1) << 1) | (gxq & 1)
    # DMA side: unit U -> (li, hy, hx, slot) or None
    image = {}
    for k in range(NDMA):
        for wv in range(12):
            for lane in range(64):
                U = (k * 12 + wv) * 64 + lane
                G, u = U >> 4, U & 15
                gx = G % GW; tmp = G // GW; hy = tmp % HH; li = tmp // HH
                s = u ^ key(li, hy >> 2, gx)
                hx = gx * 4 + (s >> 2); slot = s & 3
                ok = G < NG and hx < HW
                image[U * 16] = (li, hy, hx, slot) if ok else None
    # read side
    bank_ok = True
    for xi in range(6):
        rows = (0, 2, 4) if xi in (0, 5) else (1, 2, 3, 4)
        for r in rows:
            for c in range(6):
                addrs = []
                for lane in range(64):
                    t, kq = lane & 15, lane >> 4
                    tli = t >> (THL2 + TWL2); tty = (t >> TWL2) & ((1 << THL2) - 1); ttx = t & ((1 << TWL2) - 1)
                    G0 = (tli * HH + 4 * tty) * GW + ttx + (GW if xi == 5 else 0)
                    dr, dc, cq = r >> 2, c >> 2, c & 3
                    K = key(tli, tty + dr, ttx + dc)
                    A = G0 * 256 + ((((cq << 2) | kq) ^ K) << 4)
                    addr = A + (r * GW + dc) * 256
                    rr = r + (1 if xi == 5 else 0)
                    want = (tli, 4 * tty + rr, 4 * ttx + c, kq)
                    assert image.get(addr) == want, (GW, xi, r, c, lane, addr, image.get(addr), want)
                    addrs.append(addr)
                # ds_read_b64: lanes 0-31 / 32-63 in one LDS cycle each; bank pair = (addr/8) % 32
                for half in (0, 1):
                    banks = [(a // 8) % 32 for a in addrs[half * 32: half * 32 + 32]]
                    worst = max(banks.count(b) for b in set(banks))
                    if worst > 2: bank_ok = False
    print('GW', GW, 'NG', NG, 'NDMA', NDMA, 'image consistent; b64 conflicts <= 2-way:', bank_ok)
run(5); run(3)
