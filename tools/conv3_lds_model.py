"""LDS bank model of conv3's A-fragment reads (no GPU needed): LDS cycles per 16-lane ds_read_b128 group, averaged over the nine taps,
both wave rows, the eight row tiles and both k halves, for
  rows    the image rows in order, 128 B each, 16-byte chunk c of row r at slot c ^ (r & 7)   (conv_valid_tile<.., false>)
  planes  the label layout of Conv3Tables (csrc/az_net.hip)                                 (conv_valid_tile<.., true>)
Bank rule and lane groups: MI355X_MICROARCH.md, LDS (ds_read_b128: bank = (addr / 4) % 64, four non-contiguous groups of 16 lanes, one
extra cycle per extra distinct address on a 16-byte slot).  Measured counterpart: profiles/r03_conv3_lds_conflicts.json.
python tools/conv3_lds_model.py"""
G0 = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
G1 = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
GROUPS = [G0, G1, [l + 32 for l in G0], [l + 32 for l in G1]]
NB, IH, IW, OW, OUT_PER, IN_PER = 12, 6, 7, 5, 20, 42


def planes_cells():
    nxt, cell = [0] * 4, {}
    for bl in range(NB):
        for y in range(IH):
            for x in range(IW):
                label = (OUT_PER * bl + OW * y + x) & 15
                rho = 4 * nxt[label >> 2] + (label >> 2)
                nxt[label >> 2] += 1
                assert rho < 512
                cell[bl * IN_PER + y * IW + x] = (rho, label & 3)
    return cell


def cost(addr_of, planes):
    tot = n = 0
    for wr in (0, 1):
        for mt in range(8):
            for tap in range(9):
                for ks in (0, 1):
                    for g in GROUPS:
                        slots = {}
                        for l in g:
                            frow, fq = l & 15, l >> 4
                            ml = wr * 128 + mt * 16 + frow
                            if ml >= NB * OUT_PER:
                                ml = frow if planes else 0
                            bl, p = divmod(ml, OUT_PER)
                            r = bl * IN_PER + (p // OW + tap // 3) * IW + p % OW + tap % 3
                            a = addr_of(r, ks * 4 + fq)
                            slots.setdefault((a // 16) % 16, set()).add(a)
                        tot += max(len(v) for v in slots.values())
                        n += 1
    return tot / n


if __name__ == "__main__":
    cells = planes_cells()
    a_rows = cost(lambda r, c: r * 128 + ((c ^ (r & 7)) << 4), False)
    a_planes = cost(lambda r, c: (c & 1) * 32768 + cells[r][0] * 64 + (((c >> 1) ^ cells[r][1]) << 4), True)
    for name, a in (("rows", a_rows), ("planes", a_planes)):
        total = 16 * a + 8 * 1.0            # per K-step and wave: 16 A reads, 8 weight reads (conflict-free in both layouts)
        print(f"{name:7s} A fragment: {a:.3f} LDS cycles per lane group; conflict share of the K-step's LDS cycles {1 - 24 / total:.3f}")
