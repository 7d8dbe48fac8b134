"""Shared by tools/pmc_summary.py and tools/sq_summary.py: reading rocprofv3 CSV output of the bench command and
selecting the step-kernel dispatches of the TIMED window (the last 150 turns of the run; the pre-roll, settle and
warm-up launches before them are not part of any reported figure)."""
import csv, glob, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel-name tail, turns per launch, step launches in the timed window
FORMS = {"persistent": ("true, false>", 150, 1), "perturn": ("false, false>", 1, 150),
         "caller": ("false, false>", 1, 150),    # caller-supplied orders: per turn the action kernel(s) + the single-turn step kernel
         # learner seat: per turn evg_random_actions_seat + the one-seat instantiation of the single-turn step kernel (evg_step_vs_policy)
         "learner": ("false, false>", 1, 150)}
FORM_KEY = {"persistent": "persistent", "perturn": "one_launch_per_turn", "caller": "caller_actions_per_turn", "learner": "learner_vs_bot_per_turn"}
TWO_KERNEL_FORMS = ("caller", "learner")
ACTION_KERNELS = ("evg_random_actions_kernel", "evg_scripted_actions_kernel")
KERNELS_PER_TURN = 1            # of the caller form: set by the summary scripts (2 for random orders, 3 for two scripted agents)


def kernel_source_hash():
    """the one definition: everglades_amd._lib.kernel_source_hash (what bench.py compares with)"""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import everglades_amd
    return everglades_amd._lib.kernel_source_hash()


OBS_CTYPE = {"float32": "float", "float64": "double", "int16": "short"}
OBS_DTYPE = "float32"          # set by the summary scripts from their command line (the observation type of the profiled run)


def step_kernel(name, form):
    """the step kernel of a launch form: the two-lane kernel evg_step_kernel<OT, 64, MULTI, false> or the four-lane kernel
    evg_step4_kernel<OT, MULTI, WPE> (small batches), whichever the run used"""
    dtype = OBS_CTYPE[OBS_DTYPE]
    multi = FORMS[form][0].split(",")[0]
    if form in TWO_KERNEL_FORMS and any(k in name for k in ACTION_KERNELS):
        return True
    import re
    # evg_step_kernel<OT, 64, MULTI, MT, CHUNKED, SEAT>: the keyed-draw instantiations (MT = false) of the form's MULTI, plain or chunked; the learner form
    # runs SEAT = true
    if form == "learner":
        return re.search(r"evg_step_kernel<%s, 64, false, false, false, true(, 1)?>" % dtype, name) is not None
    # (round 5: a seventh template parameter, wavefronts per workgroup -- 1 in every product instantiation)
    if re.search(r"evg_step_kernel<%s, 64, %s, false, (true|false), false(, 1)?>" % (dtype, multi), name):
        return True
    return "evg_step4_kernel<%s, %s" % (dtype, multi) in name


def counter_rows(directory, form):
    """{counter: [value per dispatch, in dispatch order]} and the matching [duration ns] of the step kernel of `form`."""
    f = glob.glob(os.path.join(directory, "*", "*_counter_collection.csv"))[0]
    per = {}
    meta = {}
    for r in csv.DictReader(open(f)):
        if not step_kernel(r["Kernel_Name"], form):
            continue
        did = int(r["Dispatch_Id"])
        per.setdefault(did, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        meta[did] = dict(ns=int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), grid=int(r["Grid_Size"]), vgpr=int(r["VGPR_Count"]),
                         agpr=int(r["Accum_VGPR_Count"]),
                         sgpr=int(r["SGPR_Count"]), lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]), name=r["Kernel_Name"])
    ids = sorted(per)
    return [per[i] for i in ids], [meta[i] for i in ids]


def timed_window(rows, form):
    """the dispatches of the timed window: the last 150 turns of the run (one step launch each; in the caller form also the action
    kernel(s) of the turn, KERNELS_PER_TURN dispatches per turn in all)"""
    return rows[-FORMS[form][2] * (kernels_per_turn(form)):]


def kernels_per_turn(form):
    return 2 if form == "learner" else (KERNELS_PER_TURN if form == "caller" else 1)


def trace_durations(directory, form):
    """durations (ns) of the step kernel's dispatches, in dispatch order, from a --kernel-trace CSV"""
    f = glob.glob(os.path.join(directory, "*", "*_kernel_trace.csv"))[0]
    rows = [(int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if step_kernel(r["Kernel_Name"],
                                                                                                                                         form)]
    return [d for _, d in sorted(rows)]
