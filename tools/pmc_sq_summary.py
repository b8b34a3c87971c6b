#!/usr/bin/env python3
"""Per-kernel SQ counter ratios from a `rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY
SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace` run.
    python tools/pmc_sq_summary.py <counter_collection.csv>
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); the wait/active columns are fractions
of SQ_WAVE_CYCLES."""
import collections, csv, sys
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("s3d::", "").replace("void ", "").split("(")[0][:70]
    a = agg.setdefault(name, collections.Counter())
    a[r["Counter_Name"]] += float(r["Counter_Value"])
for name, c in agg.items():
    wc = max(c["SQ_WAVE_CYCLES"], 1.0)
    gui = max(c["GRBM_GUI_ACTIVE"], 1.0)
    print(f"{name:72s} mfma_busy={c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8 * 1024):6.3f} wait_any={c['SQ_WAIT_ANY'] / wc:5.3f} "
          f"wait_inst={c['SQ_WAIT_INST_ANY'] / wc:5.3f} active={c['SQ_ACTIVE_INST_ANY'] / wc:5.3f} wait_lds={c['SQ_WAIT_INST_LDS'] / wc:5.3f}")
