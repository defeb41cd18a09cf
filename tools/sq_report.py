#!/usr/bin/env python3
"""Per-(kernel, grid) digest of the three tools/profile_sq_deep.sh passes:  python tools/sq_report.py gpurun_out/r03b [filter]"""
import csv
import sys


def load(f):
    try:
        return {(r['kernel'], r['wg_x'], r['wg_y']): r for r in csv.DictReader(open(f))}
    except OSError:
        return {}


def main():
    base = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    A, B, Cc = load(base + '_sqdeep_a.csv'), load(base + '_sqdeep_b.csv'), load(base + '_sqdeep_c.csv')

    def g(d, k, c):
        try:
            return float(d[k][c + '_per_call'])
        except (KeyError, ValueError):
            return float('nan')
    keys = sorted(A, key=lambda k: -float(A[k]['total_ms']))
    for k in keys[:40]:
        if filt and filt not in k[0]:
            continue
        a = A[k]
        us = float(a['avg_us'])
        wc = g(A, k, 'SQ_WAVE_CYCLES')
        nm = g(B, k, 'SQ_INSTS_MFMA')
        bcu = g(Cc, k, 'SQ_BUSY_CU_CYCLES')
        gui = g(Cc, k, 'GRBM_GUI_ACTIVE')
        mf = g(A, k, 'SQ_VALU_MFMA_BUSY_CYCLES')
        if not wc:
            continue
        print(f"{k[0][:56]:56s} g={k[1]:>5s} n={a['calls']:>3s} {us:7.1f}us mfma_busy={mf / (4 * bcu) if bcu else 0:4.2f} wait_any={g(A, k, 'SQ_WAIT_ANY') / wc:4.2f} "
              f"wait_inst={g(A, k, 'SQ_WAIT_INST_ANY') / wc:4.2f} act_valu={g(A, k, 'SQ_ACTIVE_INST_VALU') / wc:4.2f} act_lds={g(A, k, 'SQ_ACTIVE_INST_LDS') / wc:4.2f} "
              f"V/M={g(B, k, 'SQ_INSTS_VALU') / nm if nm else 0:5.2f} L/M={g(B, k, 'SQ_INSTS_LDS') / nm if nm else 0:4.2f} VM/M={g(B, k, 'SQ_INSTS_VMEM_RD') / nm if nm else 0:4.2f} "
              f"S/M={g(B, k, 'SQ_INSTS_SALU') / nm if nm else 0:4.2f} wait_lds={g(B, k, 'SQ_WAIT_INST_LDS') / wc if k in B else 0:4.2f} clk={gui / 8 / us / 1e3 if gui == gui else 0:4.2f}GHz "
              f"waves/cu={wc * 4 / bcu if bcu else 0:4.1f}")


if __name__ == "__main__":
    main()
