"""Loops of one kernel in a disassembled code object (llvm-objdump -d --no-show-raw-insn): length and instruction mix of
every backward branch's body -- where AGPR moves, LDS reads and scratch traffic sit inside the hot loops.
usage: isa_loops.py <file.s> <kernel-name-substring>"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
start = next(i for i, l in enumerate(txt) if re.match(r'^[0-9a-f]+ <', l) and name in l)
end = next((i for i in range(start + 1, len(txt)) if re.match(r'^[0-9a-f]+ <', txt[i])), len(txt))
base = int(txt[start].split()[0], 16)
ins = []
for l in txt[start + 1:end]:
    m = re.match(r'\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*\S+(.*)', l)
    if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2), m.group(4)))
pos = {a: i for i, (a, _, _, _) in enumerate(ins)}
print(len(ins), "instructions")
loops = []
for i, (a, op, args, tail) in enumerate(ins):
    if op.startswith('s_cbranch') or op == 's_branch':
        m = re.search(r'\+0x([0-9a-f]+)>', tail)
        if not m: continue
        tgt = base + int(m.group(1), 16)
        if tgt <= a and tgt in pos:
            body = ins[pos[tgt]:i + 1]
            c = lambda p: sum(1 for x in body if x[1].startswith(p))
            loops.append((len(body), c('v_pk_fma'), c('v_accvgpr_read'), c('v_accvgpr_write'), c('ds_read') + c('ds_load'), c('ds_write') + c('ds_store'), c('scratch_load'), c('scratch_store'),
                          c('v_fma') + c('v_fmac'), c('v_mov_b32_dpp') + sum(1 for x in body if 'quad_perm' in x[2]), c('s_waitcnt'), hex(tgt - base)))
loops.sort(reverse=True)
print("   len pk_fma acc_rd acc_wr ds_rd ds_wr scr_ld scr_st   fma   dpp  wait start")
for l in loops[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]: print("%6d %6d %6d %6d %5d %5d %6d %6d %5d %5d %5d %s" % l)
