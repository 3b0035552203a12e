"""Summary of the SQ / TA / L2 counter passes of the fused kernel as the JSON bench.py quotes (CSV files do not travel to the
GPU box):  python tools/make_issue_pipes.py pmc_sq.csv pmc_ta.csv pmc_l2.csv out.json [tag [pairs per launch]]"""
import csv
import json
import sys


def means(path, kernel):
    with open(path) as f:
        return {r["Counter_Name"]: float(r["Mean"]) for r in csv.DictReader(f) if r["Kernel_Name"].replace("void ", "").startswith(kernel)}


def fractions(kernel):
    sq, ta, l2 = means(sys.argv[1], kernel), means(sys.argv[2], kernel), means(sys.argv[3], kernel)
    if not sq or not ta or not l2:
        return None
    return {"valu_busy": sq["SQ_ACTIVE_INST_VALU"] * 4 / (sq["GRBM_GUI_ACTIVE"] / 8 * 1024),
            "ta_busy_avg": ta["TA_BUSY_avr"] / (ta["GRBM_GUI_ACTIVE"] / 8), "ta_busy_max": ta["TA_BUSY_max"] / (ta["GRBM_GUI_ACTIVE"] / 8),
            "waves_waiting": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
            "l2_hit_rate": l2["TCC_HIT_sum"] / (l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]),
            "valu_instructions_per_wave": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"],
            "vmem_read_instructions_per_wave": sq["SQ_INSTS_VMEM_RD"] / sq["SQ_WAVES"],
            "scalar_memory_instructions_per_wave": (sq["SQ_INSTS_SMEM"] / sq["SQ_WAVES"]) if "SQ_INSTS_SMEM" in sq else None}


tag = sys.argv[5] if len(sys.argv) > 5 else "r02"
out = {"kernel": "k_ls_batch<1>", "workload": "sequence", "pairs_per_launch": int(sys.argv[6]) if len(sys.argv) > 6 else 64}
out.update(fractions("k_ls_batch<1>"))
out.update({
       # the all-rings launch of iteration 0 is a kernel of its own since the pooled search
       "first_launch": dict(kernel="k_ls_batch_first<1>", **(fractions("k_ls_batch_first<1>") or {})),
       "how": "vector ALU = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); texture-address units = TA_BUSY / "
              "(GRBM_GUI_ACTIVE / 8); waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
       "source": "profiles/%s_seq_pmc_sq.csv, %s_seq_pmc_ta.csv, %s_seq_pmc_l2.csv (separate rocprofv3 --pmc passes of bench.py's "
                 "sequence workload, tools/profile_%s.sh)" % (tag, tag, tag, tag)})
json.dump(out, open(sys.argv[4], "w"), indent=1)
