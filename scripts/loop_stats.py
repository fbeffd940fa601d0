"""Instruction mix of the barrier-carrying loops with >= 30 MFMAs (the FFN steady-state loops) of /tmp/t/megas.s
(produced by scripts/ffn_spills.sh)."""
import re
lines = open('/tmp/t/megas.s').read().split('\n')
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        body = lines[a:i + 1]
        mf = sum('v_mfma' in x for x in body)
        if mf >= 30 and any('s_barrier' in x for x in body):
            va = sum(bool(re.match(r'^\tv_(?!mfma)', x)) for x in body)
            sa = sum(bool(re.match(r'^\ts_(?!nop|waitcnt|barrier)', x)) for x in body)
            ds = sum('ds_read' in x for x in body)
            gl = sum('global_load_lds' in x for x in body)
            nop = sum(int(x.split()[1]) + 1 for x in body if x.strip().startswith('s_nop'))
            wc = sum('s_waitcnt' in x for x in body)
            tot = sum(bool(re.match(r'^\t[a-z]', x)) for x in body)
            print("loop lines %d-%d: mfma=%d valu=%d salu=%d ds_read=%d dma=%d nop_cycles=%d waitcnt=%d total=%d" %
                  (a, i, mf, va, sa, ds, gl, nop, wc, tot))
