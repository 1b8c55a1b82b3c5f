"""The ONE route decision (aacg_pick_route, aac.js_amd/csrc/aacg_routes.cpp): which launches a planned batch takes.  launch_run
executes the descriptor that function returns and aacg_plan_kernels prints the same descriptor, so what bench.py reports as
`roofline.kernel` is what was launched; here the decision is walked over synthetic flag combinations without a device
(aacg_debug_route builds an aacg_plan_host from flags and calls the very same function).  What every route computes:
reference src/decoder.js:218-248 + src/filter_bank.js:88-204."""
import itertools

import pytest

import aacgpu as A

Q, F = A.INPUT_QUANT_I16, A.INPUT_SPEC_F32
O32, O16 = A.OUTPUT_F32, A.OUTPUT_I16


def compose(key):
    """aacg_run_kernel_name: the symbol a set of switches stands for"""
    s = "aacg_imdct_run_" + ("quant" if key & A.RK_QUANT else "f32")
    for bit, suffix in ((A.RK_EX, "_ex"), (A.RK_DD, "_dd"), (A.RK_CPL, "_cpl"), (A.RK_RV, "_rv"), (A.RK_I16, "_i16"), (A.RK_NT, "_nt")):
        if key & bit:
            s += suffix
    return s


def test_every_registered_kernel_carries_the_symbol_its_switches_compose(engine_lib):
    reg = A.run_kernels()
    assert len(reg) == 28 and len(set(reg.values())) == 28          # one kernel per set of switches
    for name, key in reg.items():
        assert name == compose(key), (name, key)
    both = {n.replace("_quant", "_X").replace("_f32", "_X") for n in reg}
    assert len(both) == 14                                          # every variant exists for both seams


def test_named_variants_are_the_routes_of_the_batches_they_were_built_for(engine_lib):
    r = A.debug_route
    W, L, FULL = A.ROUTE_PLAN_WIDE_FRAMES, A.ROUTE_PLAN_LONG_CHAINS, A.ROUTE_PLAN_FULL_LATER_RUNS
    CI, CD, T, P = A.ROUTE_PLAN_CCE_INDEPENDENT, A.ROUTE_PLAN_CCE_DEPENDENT, A.ROUTE_PLAN_TNS, A.ROUTE_PLAN_PNS
    # plain batches: serial launches of short chains take the plain kernels, long chains the rendezvous kernels ...
    assert r(Q, O32, 0) == "aacg_imdct_run_quant" and r(F, O32, 0) == "aacg_imdct_run_f32"
    assert r(Q, O32, L) == "aacg_imdct_run_quant_rv" and r(F, O32, L | FULL) == "aacg_imdct_run_f32_rv"
    assert r(F, O32, W) == "aacg_imdct_run_f32_nt" and r(F, O32, W | L) == "aacg_imdct_run_f32_rv_nt"
    # ... unless the old route is asked for (double duty where a later run is full)
    assert r(Q, O32, L | FULL, debug_flags=A.DEBUG_ROUTE_RECOMPUTE) == "aacg_imdct_run_quant_dd"
    assert r(Q, O32, L, debug_flags=A.DEBUG_ROUTE_RECOMPUTE) == "aacg_imdct_run_quant"
    # ... and every plain batch that comes through the pipeline takes the rendezvous kernels (its launches may overlap)
    assert r(Q, O32, 0, pipelined=True) == "aacg_imdct_run_quant_rv" and r(F, O32, W, pipelined=True) == "aacg_imdct_run_f32_rv_nt"
    assert r(Q, O32, L | FULL, pipelined=True, debug_flags=A.DEBUG_ROUTE_RECOMPUTE) == "aacg_imdct_run_quant_rv"
    # int16 PCM
    assert r(Q, O16, 0) == "aacg_imdct_run_quant_i16" and r(Q, O16, W) == "aacg_imdct_run_quant_i16_nt"
    assert r(F, O16, L | FULL) == "aacg_imdct_run_f32_rv_i16" and r(F, O16, L | FULL | W) == "aacg_imdct_run_f32_rv_i16_nt"
    assert r(F, O16, L | FULL, debug_flags=A.DEBUG_ROUTE_RECOMPUTE) == "aacg_imdct_run_f32_dd_i16" and r(F, O16, L | FULL | W, debug_flags=A.DEBUG_ROUTE_RECOMPUTE) == "aacg_imdct_run_f32_dd_i16"
    assert r(Q, O16, 0, pipelined=True) == "aacg_imdct_run_quant_rv_i16" and r(F, O16, W, pipelined=True) == "aacg_imdct_run_f32_rv_i16_nt"
    # optional stages: inside the run kernel where they can be, a launch of their own otherwise
    assert r(Q, O32, T) == "aacg_imdct_run_quant_ex" and r(F, O32, T | W) == "aacg_imdct_run_f32_ex" and r(Q, O32, P) == "aacg_imdct_run_quant_ex"
    assert r(F, O32, P) == "aacg_imdct_run_f32"                                   # noise bands are a matter of the quantised seam
    assert r(Q, O16, T) == "aacg_spectral_ex_quant + aacg_imdct_run_f32_i16"
    assert r(Q, O32, T | L | FULL) == "aacg_imdct_run_quant_ex_rv"                 # a chain longer than a run: the rendezvous cut, one launch
    assert r(Q, O32, T | L | FULL, debug_flags=A.DEBUG_ROUTE_RECOMPUTE) == "aacg_spectral_ex_quant + aacg_imdct_run_f32_dd"
    assert r(Q, O32, T, pipelined=True) == "aacg_imdct_run_quant_ex_rv" and r(F, O32, T | W, pipelined=True) == "aacg_imdct_run_f32_ex_rv"
    # coupling
    assert r(Q, O32, CI) == "aacg_imdct_run_quant (coupling elements) + aacg_imdct_run_quant_cpl"
    assert r(Q, O32, CI | W) == "aacg_imdct_run_quant (coupling elements) + aacg_imdct_run_quant_cpl_nt"
    assert r(F, O32, CI | W, debug_flags=A.DEBUG_ROUTE_UNFUSED_COUPLING) == "aacg_imdct_run_f32_nt + aacg_imdct_run_f32 (coupling elements) + aacg_couple_pcm"
    assert r(Q, O32, CI | T) == "aacg_spectral_ex_quant + aacg_imdct_run_f32 (coupling elements) + aacg_imdct_run_f32_cpl"
    assert r(Q, O32, CD | T) == "aacg_spectral_ex_quant + aacg_couple_spec + aacg_spectral_ex_f32 + aacg_imdct_run_f32 + aacg_couple_pcm"
    assert r(F, O32, CD) == "copy + aacg_couple_spec + aacg_imdct_run_f32 + aacg_couple_pcm"


def test_every_combination_of_flags_has_a_registered_kernel(engine_lib):
    """no flag combination leads to a route without a kernel (aacg_debug_route refuses those), and the pipeline's choice differs
    from the serial one exactly for plain batches"""
    reg = A.run_kernels()
    flags = [A.ROUTE_PLAN_TNS, A.ROUTE_PLAN_PNS, A.ROUTE_PLAN_LONG_CHAINS, A.ROUTE_PLAN_FULL_LATER_RUNS, A.ROUTE_PLAN_WIDE_FRAMES,
             A.ROUTE_PLAN_CCE_INDEPENDENT, A.ROUTE_PLAN_CCE_DEPENDENT, A.ROUTE_PLAN_NO_RUNS]
    n = 0
    for kin, kout, dbg in itertools.product((Q, F), (O32, O16), (0, A.DEBUG_ROUTE_UNFUSED_COUPLING, A.DEBUG_ROUTE_RECOMPUTE)):
        for mask in range(1 << len(flags)):
            pf = sum(f for i, f in enumerate(flags) if mask >> i & 1)
            if kout == O16 and pf & (A.ROUTE_PLAN_CCE_INDEPENDENT | A.ROUTE_PLAN_CCE_DEPENDENT):
                continue                                            # aacg_create refuses coupling with int16 PCM
            serial, piped = A.debug_route(kin, kout, pf, False, dbg), A.debug_route(kin, kout, pf, True, dbg)
            for route in (serial, piped):
                for k in route.split(" + "):
                    k = k.replace(" (coupling elements)", "")
                    if not k and pf & A.ROUTE_PLAN_NO_RUNS:
                        continue                                    # nothing to launch
                    assert k in reg or k in ("copy", "aacg_couple_spec", "aacg_couple_pcm", "aacg_spectral_ex_quant", "aacg_spectral_ex_f32"), route
            stages = pf & A.ROUTE_PLAN_TNS or (kin == Q and pf & A.ROUTE_PLAN_PNS)
            plain = (not stages or kout == O32) and not pf & (A.ROUTE_PLAN_CCE_INDEPENDENT | A.ROUTE_PLAN_CCE_DEPENDENT | A.ROUTE_PLAN_NO_RUNS)
            assert ("_rv" in piped) == bool(plain), (kin, kout, pf, piped)
            if not plain:
                assert piped == serial
            n += 1
    assert n > 1000


@pytest.mark.parametrize("S", [2, 3])
def test_pipelined_launches_stay_behind_the_launches_whose_buffers_they_reuse(S):
    """aacg_pipeline_order (aacg_routes.cpp) is the one place the ordering of pipelined launches lives: the engine issues by it,
    the emulator schedules by it.  Launch n reads overlap buffer n mod K and writes n + 1 mod K (K = AACG_OV_BUFFERS), so it
    shares buffers and cells with launches n - K + 1, n - K, n - K - 1 (and their multiples): it has to START BEHIND every
    launch up to n - K + 1 — through its stream (in order behind its own earlier launches) and through what the host has seen
    complete before it enqueued it (the events of a marked round: a stream's launch complete means its earlier ones are)."""
    N = 240
    order = [A.pipeline_order(n, S) for n in range(N)]
    K = order[0][4]
    assert S <= A.PIPE_STREAMS
    assert K >= 3 and all(o[4] == K for o in order)
    assert {o[0] for o in order} == set(range(S))
    known = -1                                              # everything up to this launch is complete, as far as the host knows
    for n, (st, sync_round, marked, upto, _) in enumerate(order):
        assert st == n % S
        if sync_round >= 0:
            waited = [m for m in range(sync_round * S, sync_round * S + S)]
            assert all(order[m][2] for m in waited), "the host waits for round %d, whose launches carry no events" % sync_round
            assert {order[m][0] for m in waited} == set(range(S)) and max(waited) < n
            known = max(waited)                             # one launch per stream complete: everything before them too
        assert upto == known, "launch %d: the rule says everything up to %d is complete, the waits so far give %d" % (n, upto, known)
        behind = max(known, n - S if n >= S else -1)        # plus its own stream's earlier launches, which do not help with the others
        assert known >= n - (K - 1), "launch %d may start before launch %d is complete, whose buffers it shares (K = %d)" % (n, n - K + 1, K)
        assert behind < n - 1 or n < 2, "a launch must be able to run beside the one before it"
    # the streams are worth having: a launch never has to wait for the S - 1 launches before it
    assert all(order[n][3] < n - (S - 1) for n in range(S, N))
    # and the GPU's queues never run dry when the host comes back from a wait: at least a round is enqueued beyond what it waited for
    assert all(n - (order[n][1] * S + S - 1) > S for n in range(N) if order[n][1] >= 0)
