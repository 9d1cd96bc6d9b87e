"""Round 5, the sampler fault's writer: one leg of tools/sampler_repro.py per PROCESS (a StepCompany builds 32 engines on
a host thread; several in one process once hung), one JSON line per leg appended to the log.

    python tools/r05_sampler_fault.py <leg> <rounds> <log> [seconds]

legs:  lds<KiB>        the un-fenced product kernel with a dynamic-LDS request of <KiB> (0 = the 76 KiB it uses, 84, 120, 156) beside 32 lanes taking
                       rider steps (+ their prefills): does a request that keeps a SECOND SAMPLER workgroup off the CU, but admits every attention /
                       finishing workgroup, stop the fault?
       one_wg          one sampler workgroup per launch, 76 KiB
       dbg             the checking sampler at 76 KiB: the 64 wrong words and where they come from
       steps / prefill the company cut down: rider steps only / batched prefills only (76 KiB)
       alone           no company (76 KiB)
       wave / wave4    the product's one-wave sampler beside the same company (one / four streams)
       barrier<T>[_alone]   dd_tools_barrier_probe with T threads per workgroup
       pvprobe         round 4's P.V probe alone and beside 64-row GEMVs
       pkwar           dd_tools_pk_war_probe alone and beside 64-row slice GEMVs
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import sampler_repro as SR
from dropoutdecoding_amd import _lib


def pk_war(rounds, lib, beside, rows=64):
    err = torch.zeros(8, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.time()

    def body():
        for _ in range(rounds):
            rc = lib.dd_tools_pk_war_probe(16, 3072, 16, err.data_ptr(), st.cuda_stream)
            assert rc == 0, lib.dd_last_error()
            st.synchronize()

    if beside:
        co = SR.Company(lib)
        co.rows = rows
        with co:
            body()
    else:
        body()
    e = err.tolist()
    return {"test": "pk_fma_war_probe", "beside_gemvs_of_rows": rows if beside else 0, "results_checked_per_variant": rounds * 16 * 3072 * 256 * 16 * 2,
            "wrong_v0_reload_behind_packed_ops": e[0], "wrong_v1_sixteen_wait_states": e[1], "wrong_v2_distinct_registers": e[2],
            "wrong_v3_compilers_group": e[3], "wrong_v4_compilers_group_sixteen_wait_states": e[4], "wrong_v5_compilers_group_without_v_mov": e[5],
            "seconds": round(time.time() - t0, 1)}


def main():
    leg, rounds, log = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    torch.cuda.set_device(0)
    lib = _lib.load_tools()
    outs = []
    if leg.startswith("lds"):
        outs.append(SR.sampler_streams(1, rounds, lib, 32, lds_kib=int(leg[3:]), seconds=seconds))
    elif leg == "one_wg":
        outs.append(SR.sampler_streams(1, rounds, lib, 32, n_seq=1, lds_kib=0, seconds=seconds))
    elif leg == "dbg":
        outs.append(SR.sampler_streams(1, rounds, lib, 32, lds_kib=0, dbg=True, dump_prefix=os.path.splitext(log)[0] + "_dump", seconds=seconds))
    elif leg in ("steps", "prefill"):
        outs.append(SR.sampler_streams(1, rounds, lib, 32, lds_kib=0, company_mode=leg, seconds=seconds))
    elif leg == "wave":          # the product's one-wave sampler in the company that breaks the 1,024-thread forms
        outs.append(SR.sampler_streams(1, rounds, lib, 32, form="wave", seconds=seconds))
    elif leg == "wave4":         # ... from four streams at once
        outs.append(SR.sampler_streams(4, rounds, lib, 32, form="wave", seconds=seconds))
    elif leg.startswith("barrier"):      # barrier1024 / barrier512 / barrier256 [+ "_alone"]
        th = int(leg.split("_")[0][7:])
        outs.append(SR.barrier_probe(1000000, lib, 0 if leg.endswith("_alone") else 32, threads=th, seconds=seconds))
    elif leg == "pvprobe":       # round 4's P.V probe (the compiler's code) as the positive control of the write-after-read probe
        for beside, rows in ((False, 0), (True, 64)):
            outs.append(SR.pk_probe(beside, 40, lib, rows=rows or 64, pv=True))
    elif leg == "alone":
        outs.append(SR.sampler_streams(1, rounds, lib, 0, lds_kib=0, seconds=seconds))
    elif leg == "dbg_alone":
        outs.append(SR.sampler_streams(1, rounds, lib, 0, lds_kib=0, dbg=True, seconds=seconds))
    elif leg == "pkwar":
        outs.append(pk_war(rounds, lib, False))
        for rows in (64, 32, 16):
            outs.append(pk_war(rounds, lib, True, rows))
    else:
        raise SystemExit(f"unknown leg {leg}")
    with open(log, "a") as f:
        for o in outs:
            o["leg"] = leg
            line = json.dumps(o)
            print(line, flush=True)
            f.write(line + "\n")


if __name__ == "__main__":
    main()
