#!/usr/bin/env python3
"""Turns the two tools/pmc_kernel.sh logs of a kernel (SQ wave-state counters; SQ instruction / LDS counters) into the
JSON committed under profiles/:  make_sq_counters.py <log_a> <log_b> <kernel-substring> <out.json> "<note>"."""
import json
import re
import sys


def parse(path, want):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        if cur and want in cur:
            m = re.match(r"\s+(\S+)\s+([0-9.e+]+)\s+\(n=(\d+)\)", line)
            if m:
                out[m.group(1)] = float(m.group(2))
                out.setdefault("_launches", int(m.group(3)))
    return out


def main():
    a, b, want, dst, note = sys.argv[1:6]
    raw = parse(a, want)
    raw.update(parse(b, want))
    cyc = raw["GRBM_GUI_ACTIVE"] / 8.0                       # the counter is summed over the 8 XCDs
    wc = raw["SQ_WAVE_CYCLES"]
    derived = {
        "launch_cycles": cyc,
        "launch_us_at_2.4GHz": cyc / 2400.0,
        "valu_instructions_per_simd": raw["SQ_INSTS_VALU"] / 1024.0,
        "valu_cycles_per_instruction_if_never_idle": cyc / (raw["SQ_INSTS_VALU"] / 1024.0),
        "lds_instructions_per_cu": raw["SQ_INSTS_LDS"] / 256.0,
        "lds_busy_fraction": raw["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc,
        "lds_bank_conflict_share_of_lds_cycles": raw["SQ_LDS_BANK_CONFLICT"] / raw["SQ_LDS_IDX_ACTIVE"],
        "wave_cycle_breakdown": {
            "waiting_on_waitcnt_or_barrier": raw["SQ_WAIT_ANY"] / wc,
            "issue_stalled": raw["SQ_WAIT_INST_ANY"] / wc,
            "issuing": raw["SQ_ACTIVE_INST_ANY"] / wc,
        },
        "waves": raw["SQ_WAVES"],
    }
    json.dump({"kernel": want, "note": note,
               "command": "tools/pmc_kernel.sh (rocprofv3 --pmc, two passes of 8 counters, bench.py --steps 1 --sweeps-per-step 24); "
                          "means over the launches of the run; SQ_*_CYCLES are in units of 4 clocks; kernels run ~5 % slower "
                          "under counter collection",
               "raw": raw, "derived": derived}, open(dst, "w"), indent=1)
    print(json.dumps(derived, indent=1))


if __name__ == "__main__":
    main()
