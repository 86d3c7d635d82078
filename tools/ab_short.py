#!/usr/bin/env python3
"""Filter for tools/ab_hess.py output: one short line per configuration (step time, the main groups, the checksums)."""
import json, re, sys
for line in sys.stdin:
    m = re.match(r"^\[(.*?)\] (\{.*\})\s*$", line)
    if not m:
        print(line.rstrip()[:300]); continue
    d = json.loads(m.group(2))
    print(f"{m.group(1)[-44:]:44s} step {d['ms_per_step']:7.3f}  walk {d.get('vesselness', 0):6.3f}  resolve {d.get('vesselness_resolve', 0):5.3f}  "
          f"gz {d.get('gauss_z', 0):5.3f}  gyx {d.get('gauss_yx', 0):5.3f}  label {d.get('label', 0):5.3f}  crc {d['crc_frangi']:x}/{d['crc_labels']:x}")
