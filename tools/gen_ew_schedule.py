#!/usr/bin/env python
"""Generator of the gate-math schedule of rd_lstm_mfma_f16x3_t32_kernel (EW_CELL / EW_STAGE in
ribodetector_amd/csrc/rd_lstm_t32.hpp): 16 cells per lane and phase, 13 pipeline stages per cell (rd_ew_unit), cell c enters
the pipeline at step floor(6.5 c) so that two cells are in flight and never in the same stage (a dependent instruction is
always separated from its producer by the other cell's unit); stage 13 (LDS stores of a finished row-tile of 4 cells) follows
the step in which the row-tile's last cell leaves stage 12.

    python tools/gen_ew_schedule.py            # prints the two tables
    python tools/gen_ew_schedule.py --check    # compares with the tables in the source (exit status 1 on a difference)
"""
import os
import re
import sys

CELLS, STAGES, PERIOD2 = 16, 13, 13     # a new cell every 6.5 steps


def schedule():
    start = [(PERIOD2 * c) // 2 for c in range(CELLS)]
    units = []
    for step in range(start[-1] + STAGES):
        done_tile = None
        for c in range(CELLS):
            s = step - start[c]
            if 0 <= s < STAGES:
                units.append((c, s))
                if s == STAGES - 1 and c % 4 == 3:
                    done_tile = c
        if done_tile is not None:
            units.append((done_tile, 13))
    return units


def main():
    units = schedule()
    cell = ",".join(str(c) for c, _ in units)
    stage = ",".join(str(s) for _, s in units)
    if "--check" in sys.argv:
        src = open(os.path.join(os.path.dirname(__file__), "..", "ribodetector_amd", "csrc", "rd_lstm_t32.hpp")).read()
        have_c = re.search(r"EW_CELL\[EW_NU\] = \{([^}]*)\}", src).group(1).replace(" ", "")
        have_s = re.search(r"EW_STAGE\[EW_NU\] = \{([^}]*)\}", src).group(1).replace(" ", "")
        nu = int(re.search(r"constexpr int EW_NU = (\d+);", src).group(1))
        ok = have_c == cell and have_s == stage and nu == len(units)
        print("schedule in rd_lstm_t32.hpp %s the generator (%d units)" % ("matches" if ok else "DIFFERS from", len(units)))
        sys.exit(0 if ok else 1)
    print("constexpr int EW_NU = %d;" % len(units))
    print("constexpr unsigned char EW_CELL[EW_NU] = {%s};" % cell)
    print("constexpr unsigned char EW_STAGE[EW_NU] = {%s};" % stage)


if __name__ == "__main__":
    main()
