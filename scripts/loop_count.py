"""Instruction mix of every loop (label .. backward branch) of one kernel in a hipcc -S dump.
usage: python scripts/loop_count.py file.s kernel_name_substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\w*' + re.escape(key) + r'\w*:', l))
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
body = s[start:end]
print(body[0], len(body), 'lines')
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: labels[mm.group(1)] = i
def mix(seg):
    c = Counter()
    for x in seg:
        op = x.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt')): c['trans'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('s_waitcnt'): c['wait'] += 1
        elif op.startswith('s_'): c['salu'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith(('global', 'buffer')): c['vmem'] += 1
        elif op.startswith('scratch'): c['scratch'] += 1
    return dict(c)
for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.search(r's_branch\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        a = labels[mm.group(1)]
        seg = [x.strip() for x in body[a:i] if x.strip() and not x.strip().startswith(('.', ';'))]
        print(f"loop {mm.group(1)} lines {a}-{i}: {len(seg)} instrs", mix(seg))
if len(sys.argv) > 3:        # histogram of one loop: label as third argument
    lab = sys.argv[3]
    a = labels[lab]
    b = max(i for i, l in enumerate(body) if re.search(r's_c?branch\w*\s+' + re.escape(lab) + r'\b', l))
    h = Counter(x.split()[0] for x in (y.strip() for y in body[a:b]) if x and not x.startswith(('.', ';')))
    for op, n in h.most_common(40): print(f"  {n:4d} {op}")
