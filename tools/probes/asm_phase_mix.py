#!/usr/bin/env python3
"""Instruction mix between the NM_LANE_PROF marks of one kernel's assembly: asm_phase_mix.py <kernel.s> <marks.txt (line slot)>"""
import re, sys
marks = [tuple(map(int, l.split())) for l in open(sys.argv[2])]
lines = open(sys.argv[1]).read().split('\n')
PI = re.compile(r'^\s+[vsbdg][a-z]*_')
prev = marks[0][0]
for (ln, slot) in marks[1:]:
    seg = lines[prev:ln]
    c = lambda pat: sum(1 for x in seg if re.search(pat, x))
    print("lines %d-%d -> slot %d: instr %d f64 %d bl %d bs %d sl %d ss %d gl %d gs %d ds %d accr %d accw %d vmwait %d br %d" % (
        prev, ln, slot, sum(1 for x in seg if PI.search(x)), c('_f64'), c('buffer_load'), c('buffer_store'), c('scratch_load'), c('scratch_store'),
        c('global_load'), c('global_store'), c(r'\bds_'), c('accvgpr_read'), c('accvgpr_write'), c('s_waitcnt vmcnt'), c('s_cbranch')))
    prev = ln
