#!/usr/bin/env python3
"""Compact per-kernel resource + instruction-mix report from build/yolo2_hip.s (make asm)."""
import re, sys, collections
path = sys.argv[1] if len(sys.argv) > 1 else "build/yolo2_hip.s"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
txt = open(path).read()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt and flt not in name: continue
    ops = collections.Counter(l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";")))
    md = re.search(r"\.name:\s+%s\n(.*?)\.wavefront_size" % re.escape(name), txt, re.S)
    vg = re.search(r"\.vgpr_count:\s+(\d+)", md.group(1)).group(1) if md else "?"
    sg = re.search(r"\.sgpr_count:\s+(\d+)", md.group(1)).group(1) if md else "?"
    key = lambda k: sum(v for o, v in ops.items() if o.startswith(k))
    print(f"{name[:70]:70s} vgpr={vg} sgpr={sg} dot2={key('v_dot2')} med3={key('v_med3')} ashr={key('v_ashr')} add={key('v_add')} mov={key('v_mov')} "
          f"sload={key('s_load')} ds_r={key('ds_read')} ds_w={key('ds_write')} gload={key('global_load')} total_valu={sum(v for o,v in ops.items() if o.startswith('v_'))}")
