#!/usr/bin/env python3
"""One line per record of a tools/probe_patch.py log."""
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print("%-3s %-22s apply %7.1f us  steps %4d  solve %7.1f ms  numbering %6.2f  assembly %5.2f  diff %.1e" % (
            r["size"], r["op"], r["apply_us"], r["pcg_steps"], r["solve_ms"], r["symbolic_ms"], r["assemble_ms"], r["rel_diff_vs_csr"]))
    else:
        print(l.strip()[:200])
