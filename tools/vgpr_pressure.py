"""Development aid: live vector registers along the frame loop of a kernel, from the compiler's assembly
(exp/<name>/dev.s of tools/exp_build.sh).  A backward liveness scan over the loop body treated as straight-line
code (two rounds, so that loop-carried values are seen); prints the live count at barriers, priority changes,
scheduling fences and at the peak.     usage: python tools/vgpr_pressure.py dev.s <kernel name substring>"""
import re
import sys

text = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(_ZN3voc\d+' + key + r'[^\n:]*):[^\n]*\n(.*?)s_endpgm', text, re.S | re.M)
body = m.group(2).split('\n')
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
best = (0, 0, 0)
for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', l)
    if mm:
        t = mm.group(1) or mm.group(2)
        if t in labels and labels[t] < i and i - labels[t] > best[0]:
            best = (i - labels[t], labels[t], i)
_, lo, hi = best
loop = body[lo:hi + 1]


def regs(tok):
    tok = tok.strip().strip(',')
    mm = re.match(r'^-?\|?v(\d+)\|?$', tok)
    if mm:
        return [int(mm.group(1))]
    mm = re.match(r'^-?\|?v\[(\d+):(\d+)\]\|?$', tok)
    if mm:
        return list(range(int(mm.group(1)), int(mm.group(2)) + 1))
    return []


STORES = ('ds_write', 'global_store', 'scratch_store', 'buffer_store', 'flat_store')
ins = []
for l in loop:
    s = l.split(';')[0].strip()
    note = l.strip() if ('sched_barrier' in l or 's_barrier' in l or 's_setprio' in l) else None
    if not s or s.endswith(':') or s.startswith('.'):
        if note:
            ins.append((None, [], [], note))
        continue
    op, _, rest = s.partition(' ')
    ops = [o for o in re.split(r',\s*', rest) if o]
    defs, uses = [], []
    if op.startswith(STORES) or op.startswith('v_cmp') or op.startswith('s_'):
        for o in ops:
            uses += regs(o)
    else:
        if ops:
            defs = regs(ops[0])
        for o in ops[1:]:
            uses += regs(o)
        if op.startswith(('v_fmac', 'v_mac', 'v_pk_fmac')) or 'dpp' in op or op.startswith('v_swap'):
            uses += defs
    ins.append((op, defs, uses, note))
live = set()
counts = [0] * len(ins)
for rnd in range(2):
    for i in range(len(ins) - 1, -1, -1):
        op, defs, uses, note = ins[i]
        live -= set(defs)
        live |= set(uses)
        counts[i] = len(live)
peak = max(range(len(ins)), key=lambda i: counts[i])
print(f'loop of {len(ins)} entries, peak {counts[peak]} live VGPRs at entry {peak}')
for i, (op, defs, uses, note) in enumerate(ins):
    if note or i == peak:
        print(f'  {i:5d} live {counts[i]:3d}  {note or op}')
