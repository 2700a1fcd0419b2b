"""The compiled step program (stochastic_gcn_amd/step_program.py + sgcn_step_run: ONE foreign call per
training step) against the eager per-layer host path: same kernels, same arguments -> bit-identical
weights, Adam moments, history, loss and accuracy over consecutive steps, for every layer stack the
compiler supports (sparse input features included); unsupported stacks (wide LayerNorm layers) fall back."""
import numpy as np
import pytest
import torch

import model_cases as mc

pytestmark = pytest.mark.gpu


def _model(case, params, native, group=True, multitask=False):
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.vrgcn import VRGCN
    from stochastic_gcn_amd.plaingcn import PlainGCN
    FLAGS.reset()
    FLAGS.update(**{k: v for k, v in case['flags'].items() if hasattr(FLAGS, k)})
    FLAGS.update(native_step=native, batch_size=case['cfg']['batch'], group_dw=group, lean_sync=group, agg_overlap=not group)
    cls = VRGCN if case['cfg']['model'] == 'vr' else PlainGCN
    fl = case['flags']
    m = cls(fl['num_layers'], fl['preprocess'], case['ph'], case['feats'], case['nbr'], case['adj'], fl['cvd'],
            is_training=True, device=torch.device('cuda:0'), multitask=multitask)
    m.set_params(params)
    return m


def _run(case, native, steps, slot, group=True, multitask=False):
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.scheduler import StagingSlot
    params = mc.make_oracle_model(case, seed=3).params
    m = _model(case, {k: v.copy() for k, v in params.items()}, native, group, multitask)
    sch = mc.make_scheduler(case, 1)
    slots = [StagingSlot(pin=True) for _ in range(3)] if slot else None
    losses = []
    for step in range(steps):
        if sch.start >= sch.data.shape[0]:
            sch.start = 0
        pb = sch.minibatch_packed(case['cfg']['batch'], FLAGS.plan_t, slots[step % 3] if slot else None)
        pb.dropout = case['flags']['dropout']
        out = m.run_one_step(None, pb, sync=False)
        losses.append((out[1].clone(), out[2].clone()))
    torch.cuda.synchronize()
    return m, losses


SUPPORTED = ['reddit_cvd_pp', 'reddit_cv_pp', 'cvd_pp_L3', 'cv_nopp_L2', 'ns_nopp_L2', 'ns_nopp_L2_reverse', 'is_pp']


@pytest.mark.parametrize("name", SUPPORTED)
@pytest.mark.parametrize("slot,group", [(False, True), (True, True), (True, False), (False, False)])
def test_program_is_bit_identical_to_the_eager_path(name, slot, group):
    """group: the layers' weight-gradient GEMMs recorded and issued as ONE grouped launch + ONE reduction launch, gradients
    STORED (no memset), loss statistics in the optimizer's launch, history scatter after the optimizer, the aggregator fused
    (the defaults: --group_dw --lean_sync --noagg_overlap) or everything layer by layer / on the auxiliary stream -- the same
    tiles, K slices and order of additions either way."""
    case = mc.build_case(name)
    a, la = _run(case, False, 5, slot)
    b, lb = _run(case, True, 5, slot, group=group)
    progs = getattr(b, '_programs', {})
    assert progs and all(p is not None for p in progs.values()), getattr(b, '_program_note', 'no program was compiled')
    assert not getattr(a, '_programs', {})
    assert torch.equal(a.theta, b.theta) and torch.equal(a.adam_m, b.adam_m) and torch.equal(a.adam_v, b.adam_v)
    for ha, hb in zip(a.history, b.history):
        assert torch.equal(ha[0], hb[0])
    for (l1, a1), (l2, a2) in zip(la, lb):
        assert torch.equal(l1, l2) and torch.equal(a1, a2)
    assert a.dropout_step == b.dropout_step == 5 and a.adam_t == b.adam_t == 5
    assert a.amt_data == b.amt_data and np.array_equal(a.field_sizes, b.field_sizes)      # the epoch counters too
    prog = next(iter(progs.values()))
    from stochastic_gcn_amd.step_program import OP
    assert sum(1 for o, _ in prog.ops_fb if o == OP['DW_FLUSH']) == (1 if group else 0)
    assert sum(1 for o, _ in prog.ops_fb if o == OP['GRAD_STORE']) == (1 if group else 0)
    assert sum(1 for o, _ in prog.ops_fb if o == OP['AUX_MEMSET0']) == (0 if group else 1)
    # the control-variate aggregator: fused on the step's stream (default), or its history half first on the auxiliary stream
    n_agg = sum(1 for l in b.layers if type(l).__name__ == 'VRAggregator')
    assert sum(1 for o, _ in prog.ops_fb if o == OP['VR_AGG']) == (n_agg if group else 0)
    assert sum(1 for o, _ in prog.ops_fb if o == OP['VR_AGG_PRE']) == (0 if group else n_agg)
    print("%s: %d ops per step, arena %.1f MB" % (name, prog.n_all, prog.arena.numel() * 4 / 2 ** 20))


@pytest.mark.parametrize("name", sorted(mc.DET_CASES))
@pytest.mark.parametrize("slot,group", [(False, True), (True, True), (False, False)])
def test_det_dropout_stacks_run_as_programs_bit_identical_to_the_eager_path(name, slot, group):
    """--det_dropout (gcn/layers.py:141-202, 236-248, 320-349, 425-428; round 6, ABI v16): DetDropoutFC, both aggregators on
    (mean, variance) with a mean and a variance history, the medg-weighted third matrix and its transpose (its values travel in
    the packed minibatch in both orders), Gaussian re-sampling -- the eager layers' calls as ops GEMM .. GATE of the step program:
    the same weights, Adam moments, BOTH histories, losses, bit for bit."""
    case = mc.build_case(name)
    a, la = _run(case, False, 4, slot)
    b, lb = _run(case, True, 4, slot, group=group)
    progs = getattr(b, '_programs', {})
    assert progs and all(p is not None for p in progs.values()), getattr(b, '_program_note', 'no program was compiled')
    assert not getattr(a, '_programs', {})
    for (l1, a1), (l2, a2) in zip(la, lb):
        assert torch.equal(l1, l2) and torch.equal(a1, a2), (l1, l2)
    assert torch.equal(a.theta, b.theta) and torch.equal(a.adam_m, b.adam_m) and torch.equal(a.adam_v, b.adam_v)
    for ha, hb in zip(a.history, b.history):
        assert len(ha) == len(hb) and all(torch.equal(x, y) for x, y in zip(ha, hb))
    assert any(float(h.abs().sum()) > 0 for hs in b.history for h in hs) or not b.history
    prog = next(iter(progs.values()))
    from stochastic_gcn_amd.step_program import OP
    codes = [o for o, _ in prog.ops_fb]
    n_fc = sum(1 for l in b.layers if type(l).__name__ == 'DetDropoutFC')
    assert prog.det and codes.count(OP['DET_RELU_FWD']) == n_fc == codes.count(OP['DET_RELU_BWD']) and codes.count(OP['GAUSS']) == 1
    assert codes.count(OP['MEMSET0']) == 1 and codes.count(OP['GRAD_STORE']) == 0
    print("%s: %d ops per step, arena %.1f MB" % (name, prog.n_all, prog.arena.numel() * 4 / 2 ** 20))


@pytest.mark.parametrize("name", ['det_cv_pp_L3', 'det_ns_nopp_L2'])
def test_det_dropout_evaluation_model_runs_as_a_program_too(name):
    """The TEST model of a det-dropout stack (is_training False: forward, Gaussian re-sampling, loss, prediction, both test
    histories' scatter; dropout 0 -> keep 1) as a forward-only program against its layer-by-layer run: the same loss,
    accuracy, predictions and histories, bit for bit, over consecutive evaluation batches (the second reads what the first
    wrote)."""
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.vrgcn import VRGCN
    from stochastic_gcn_amd.plaingcn import PlainGCN
    case = mc.build_case(name)
    params = mc.make_oracle_model(case, seed=3).params
    res = {}
    for native in (False, True):
        FLAGS.reset()
        FLAGS.update(**{k: v for k, v in case['flags'].items() if hasattr(FLAGS, k)})
        FLAGS.update(native_step=native, batch_size=case['cfg']['batch'], test_batch_size=case['cfg']['batch'],
                     test_degree=case['flags']['degree'])
        cls = VRGCN if case['cfg']['model'] == 'vr' else PlainGCN
        fl = case['flags']
        m = cls(fl['num_layers'], fl['preprocess'], case['ph'], case['feats'], case['nbr'], case['adj'], fl['cvd'],
                is_training=False, device=torch.device('cuda:0'))
        m.set_params({k: v.copy() for k, v in params.items()})
        sch = mc.make_scheduler(case, 1)
        outs = []
        for _ in range(3):
            pb = sch.minibatch_packed(case['cfg']['batch'], FLAGS.plan_t, None)
            o = m.run_one_step(None, pb, sync=True)
            outs.append((o[0], o[1], np.array(o[2])))
        progs = [p for p in getattr(m, '_programs', {}).values()]
        assert bool(progs and all(p is not None and p.det for p in progs)) == native, getattr(m, '_program_note', None)
        res[native] = (outs, [h.clone() for hs in m.history for h in hs])
    for (l0, a0, p0), (l1, a1, p1) in zip(res[False][0], res[True][0]):
        assert l0 == l1 and a0 == a1 and np.array_equal(p0, p1)
    assert len(res[False][1]) == len(res[True][1]) and all(torch.equal(x, y) for x, y in zip(res[False][1], res[True][1]))


def test_dropout_zero_and_weight_decay_variants():
    for name, extra in (('reddit_cvd_pp', dict(dropout=0.0)), ('reddit_cv_pp', dict(weight_decay=5e-3)),
                        ('ns_nopp_L2', dict(weight_decay=1e-3, dropout=0.0))):
        case = mc.build_case(name)
        case['flags'].update(extra)
        a, la = _run(case, False, 3, False)
        b, lb = _run(case, True, 3, False)
        assert all(p is not None for p in b._programs.values()), getattr(b, '_program_note', None)
        assert torch.equal(a.theta, b.theta)
        assert all(torch.equal(x[0], y[0]) for x, y in zip(la, lb))


@pytest.mark.parametrize("name", ['pubmed_cvd_pp', 'cora_exact'])
@pytest.mark.parametrize("slot,group", [(False, True), (True, True), (False, False)])
def test_sparse_input_stacks_run_as_programs_bit_identical_to_the_eager_path(name, slot, group):
    """BASELINE configs 1 and 2 (sparse feature matrices, gcn/layers.py:125,401-402 with sparse_inputs): the slice of the
    feature CSR, the sparse dropout, the sparse products, their LayerNorm passes and the transposed product of the weight
    gradient are ops of the step program (CSR_SLICE, DROPOUT, SPMM, LN_ACT_FWD / _BWD, CSR_TRANSPOSE, GATHER_F32) -- the
    calls of the eager layers, in their order."""
    case = mc.build_case(name)
    a, la = _run(case, False, 4, slot)
    b, lb = _run(case, True, 4, slot, group=group)
    progs = getattr(b, '_programs', {})
    assert progs and all(p is not None for p in progs.values()), getattr(b, '_program_note', 'no program was compiled')
    assert torch.equal(a.theta, b.theta) and torch.equal(a.adam_m, b.adam_m) and torch.equal(a.adam_v, b.adam_v)
    for ha, hb in zip(a.history, b.history):
        assert torch.equal(ha[0], hb[0])
    for (l1, a1), (l2, a2) in zip(la, lb):
        assert torch.equal(l1, l2) and torch.equal(a1, a2)
    assert a.amt_data == b.amt_data and np.array_equal(a.field_sizes, b.field_sizes)
    prog = next(iter(progs.values()))
    from stochastic_gcn_amd.step_program import OP
    ops = [o for o, _ in prog.ops_fb]
    assert prog.sparse and ops.count(OP['CSR_SLICE']) == 1 and ops.count(OP['CSR_TRANSPOSE']) == 1
    assert ops.count(OP['MEMSET0']) == 1 and ops.count(OP['GRAD_STORE']) == 0      # sums into a zeroed gradient buffer
    print("%s: %d ops per step, arena %.1f MB, slice capacity %d nonzeros" % (name, prog.n_all, prog.arena.numel() * 4 / 2 ** 20,
                                                                              prog.nnz_cap))


@pytest.mark.parametrize("name", ['reddit_cvd_pp_wide'])
def test_unsupported_stacks_fall_back_to_the_eager_path(name):
    case = mc.build_case(name)
    b, lb = _run(case, True, 2, False)
    assert list(b._programs.values()) == [None] and b._program_note
    a, la = _run(case, False, 2, False)
    assert torch.equal(a.theta, b.theta)


def test_minibatch_that_does_not_fit_runs_eagerly():
    case = mc.build_case('reddit_cvd_pp')
    b, _ = _run(case, True, 1, False)
    prog = next(iter(b._programs.values()))
    sch = mc.make_scheduler(case, 1)
    pb = sch.batch_packed(case['train'][:case['cfg']['batch']], 0, None)
    assert prog.fits(pb)
    prog._cap_max[:] = 1                      # pretend the arena was sized for one row
    assert not prog.fits(pb)
    pb.dropout = case['flags']['dropout']
    out = b.run_one_step(None, pb, sync=True)             # falls back, still a valid step
    assert np.isfinite(out[1])


def test_unsynchronised_results_alias_the_programs_statistics_slot():
    """run_one_step(sync=False) on the program path hands out VIEWS of the program's statistics slot (documented on
    run_one_step): the next step overwrites them, so a caller that keeps per-step values clones them."""
    from stochastic_gcn_amd.flags import FLAGS
    case = mc.build_case('reddit_cvd_pp')
    params = mc.make_oracle_model(case, seed=3).params
    m = _model(case, {k: v.copy() for k, v in params.items()}, True)
    sch = mc.make_scheduler(case, 1)
    outs, kept = [], []
    for _ in range(2):
        pb = sch.minibatch_packed(case['cfg']['batch'], FLAGS.plan_t, None)
        pb.dropout = case['flags']['dropout']
        o = m.run_one_step(None, pb, sync=False)
        outs.append(o)
        kept.append(o[1].clone())
    torch.cuda.synchronize()
    assert outs[0][1].data_ptr() == outs[1][1].data_ptr()            # same slot ...
    assert float(outs[0][1]) == float(kept[1]) != float(kept[0])     # ... holding the LAST step's loss


def test_program_is_rebuilt_when_what_it_baked_in_changes():
    """A program bakes in raw addresses and whether the history update is local: a hook attached, or a
    tensor re-allocated, after the first step must compile a new program instead of running the stale one."""
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd import ops
    case = mc.build_case('reddit_cvd_pp')
    params = mc.make_oracle_model(case, seed=3).params
    m = _model(case, {k: v.copy() for k, v in params.items()}, True)
    ref = _model(case, {k: v.copy() for k, v in params.items()}, False)
    sch, sch2 = mc.make_scheduler(case, 1), mc.make_scheduler(case, 1)

    def step(model, s):
        FLAGS.update(native_step=model is m)          # FLAGS is global: the reference model runs eagerly
        pb = s.minibatch_packed(case['cfg']['batch'], FLAGS.plan_t, None)
        pb.dropout = case['flags']['dropout']
        return model.run_one_step(None, pb, sync=True)
    step(m, sch), step(ref, sch2)
    assert len(m._programs) == 1 and not getattr(ref, '_programs', None)
    calls = []

    def hook(hist, idx, rows, scatter):          # what parallel.DataParallel.sync_history does on one rank
        calls.append(int(rows.shape[0]))
        scatter(hist, idx, rows)
    m.history_hook = hook
    step(m, sch), step(ref, sch2)
    assert len(m._programs) == 2 and len(calls) == 1          # a second program, whose history update went through the hook
    for p in m._programs.values():
        p._seen = True
    m.history[0][0] = m.history[0][0].clone()                  # the history tensor moves
    step(m, sch), step(ref, sch2)
    # a third program -- and the oldest (its arena with it) has been dropped: a model keeps two
    assert len(m._programs) == 2 and sum(not getattr(p, '_seen', False) for p in m._programs.values()) == 1
    torch.cuda.synchronize()
    assert torch.equal(m.theta, ref.theta) and torch.equal(m.history[0][0], ref.history[0][0])


def test_two_programs_of_one_process_keep_their_own_stream_mode():
    """the auxiliary-stream / fusion mode is a program's own (SGCN_OP_MODE), not the process-wide knob the LAST
    built program left behind: a model compiled for the auxiliary stream (agg_overlap) and one compiled for a single stream,
    stepped alternately, each equal their own eager twin bit for bit; the knob itself is what it was."""
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd import _ffi
    from stochastic_gcn_amd.step_program import OP
    case = mc.build_case('reddit_cvd_pp')
    params = mc.make_oracle_model(case, seed=3).params
    models, refs, schs = [], [], []
    for group in (False, True):                                   # (False: overlap on the auxiliary stream; True: one stream)
        models.append(_model(case, {k: v.copy() for k, v in params.items()}, True, group))
        refs.append(_model(case, {k: v.copy() for k, v in params.items()}, False, group))
        schs.append((mc.make_scheduler(case, 1), mc.make_scheduler(case, 1)))
    knob = int(_ffi.lib.sgcn_tune_get(b"step_overlap"))
    for step in range(3):
        for i, group in enumerate((False, True)):
            for model, sch, native in ((models[i], schs[i][0], True), (refs[i], schs[i][1], False)):
                FLAGS.update(native_step=native, group_dw=group, lean_sync=group, agg_overlap=not group)
                pb = sch.minibatch_packed(case['cfg']['batch'], FLAGS.plan_t, None)
                pb.dropout = case['flags']['dropout']
                model.run_one_step(None, pb, sync=True)
    torch.cuda.synchronize()
    for m, r in zip(models, refs):
        assert torch.equal(m.theta, r.theta) and torch.equal(m.history[0][0], r.history[0][0])
    p0, p1 = (next(iter(m._programs.values())) for m in models)
    assert (p0.overlap, p1.overlap) == (1, 0)
    assert all(o == OP['MODE'] for o in (p0.c_fb[0].op, p1.c_fb[0].op))
    assert int(_ffi.lib.sgcn_tune_get(b"step_overlap")) == knob


def test_dense_scratch_is_sized_for_the_row_capacity():
    """The program's dense-layer scratch covers every op at its ROW CAPACITY (LayerNorm-backward partials grow with
    the rows; ADVICE r2: a fixed 4M-float buffer failed inside sgcn_step_run for narrow layers on large row caps)."""
    from stochastic_gcn_amd._ffi import lib
    from stochastic_gcn_amd.step_program import StepProgram, GEMM_WS_BOUND
    case = mc.build_case('reddit_cvd_pp')
    params = mc.make_oracle_model(case, seed=3).params
    m = _model(case, {k: v.copy() for k, v in params.items()}, True)
    prog = StepProgram(m, 0.2)
    h = case['flags']['hidden1']
    need = (int(lib.sgcn_ln_act_bwd_ws_floats(prog.caps[0], h)) + 3) // 4 * 4 + GEMM_WS_BOUND
    assert prog._ws_need >= need and prog.gemm_ws_floats >= prog._ws_need
    for M in (1, 31, 512, 2042, 70000):
        for N in (16, 41, 128):
            for K in (24, 128, 1204):
                assert lib.sgcn_gemm_ws_floats(M, N, K) <= GEMM_WS_BOUND and lib.sgcn_gemm_ws_floats(K, N, M) <= GEMM_WS_BOUND


@pytest.mark.parametrize("fuse", [0, 1, 2, 3, 4, 8, 15, 17, 31, 32, 63, 72, 127])
def test_output_layer_in_the_loss_kernel_or_as_its_own_launches(fuse):
    """sgcn_step_run folds the output layer into the loss kernel's row pass (its forward product as the head, its input
    gradient as the tail; a narrow dense layer in the reduce pass of the split-K layer in front of it; a layer's input
    gradient in its LayerNorm backward pass; the dense layer in front of the output layer as the pre-layer of the loss kernel's
    head; the weight gradients' reductions in the optimizer's launch; the first layer's LayerNorm backward behind the second
    layer's row pass: knob step_fuse, default 127 -- what every other test of this file runs); with the fusion partly or
    wholly off the same program issues the separate launches, and every variant gives the eager path's bits."""
    from stochastic_gcn_amd import _ffi
    # (the mid-size Reddit recipe: its first layer, 192 inputs on ~1,000 rows, is cut over K like the full-size one -- the
    # miniature cases' layers are too small to be -- so bit 2 of the knob has something to fold)
    case = mc.build_case(mc.REDDIT_MID)
    assert _ffi.lib.sgcn_gemm_ws_floats(2 * 400, 64, 192) > 0
    _ffi.tune('step_fuse', 0)            # (bit 3 acts inside sgcn_dense_bwd_f32, i.e. on the eager path too)
    try:
        a, la = _run(case, False, 4, False)
    finally:
        _ffi.tune('step_fuse', 127)
    _ffi.tune('step_fuse', fuse)
    try:
        b, lb = _run(case, True, 4, False)
    finally:
        _ffi.tune('step_fuse', 127)
    assert all(p is not None for p in b._programs.values())
    assert torch.equal(a.theta, b.theta) and torch.equal(a.adam_m, b.adam_m)
    for ha, hb in zip(a.history, b.history):
        assert torch.equal(ha[0], hb[0])
    for (l1, a1), (l2, a2) in zip(la, lb):
        assert torch.equal(l1, l2) and torch.equal(a1, a2)


def test_staged_batches_cycle_through_the_models_ring_of_device_buffers():
    """Model.stage copies a pinned minibatch into a ring of device buffers one batch ahead of its step; a buffer is written
    again a few groups of sixteen batches later (eight by default, three here), behind an event of the step's stream recorded once per group.  110 steps
    (the ring more than twice around) with the copy a batch ahead, against the eager path fed from pageable memory; a
    minibatch that is run after its buffer has been handed on is refused."""
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.scheduler import StagingSlot
    case = mc.build_case('reddit_cvd_pp')
    a, la = _run(case, False, 110, False)
    params = mc.make_oracle_model(case, seed=3).params
    b = _model(case, {k: v.copy() for k, v in params.items()}, True)
    b._RING_GROUPS = 3                        # (default 8: 128 buffers; 48 here, so that 110 steps go round more than twice)
    sch = mc.make_scheduler(case, 1)
    slots = [StagingSlot(pin=True) for _ in range(4)]

    def fetch(i):
        if sch.start >= sch.data.shape[0]:
            sch.start = 0
        pb = sch.minibatch_packed(case['cfg']['batch'], FLAGS.plan_t, slots[i % 4])
        pb.dropout = case['flags']['dropout']
        return pb
    lb, first = [], None
    nxt = fetch(0)
    for step in range(110):
        pb, nxt = nxt, fetch(step + 1)
        b.stage(nxt)
        first = first or pb
        out = b.run_one_step(None, pb, sync=False)
        lb.append((out[1].clone(), out[2].clone()))
    torch.cuda.synchronize()
    assert b._ring['n'] == 111 and len(b._ring['bufs']) == 48 and all(t is not None for t in b._ring['bufs'])
    assert torch.equal(a.theta, b.theta) and torch.equal(a.adam_m, b.adam_m)
    for ha, hb in zip(a.history, b.history):
        assert torch.equal(ha[0], hb[0])
    for (l1, a1), (l2, a2) in zip(la, lb):
        assert torch.equal(l1, l2) and torch.equal(a1, a2)
    with pytest.raises(RuntimeError, match="handed on"):
        b.run_one_step(None, first, sync=False)


def test_full_size_reddit_program_equals_the_unfolded_eager_path():
    """BASELINE config 3 at FULL size (S-Reddit: 232,965 vertices, 602 features, hidden 128, batch 512: the shapes the
    bench's epoch runs -- first layer cut over K into the split-K reduce pass, 256 x 128 and 128 x 128 weights by swizzled
    direct loads, the chained LayerNorm backward, 41 classes in the loss kernel): three steps as the step program with every
    fold on, against the eager per-layer path with none -- the path tests/test_model_gpu.py holds against the NumPy oracle
    at this size -- bit for bit."""
    from stochastic_gcn_amd import _ffi, synthetic
    from oracle import model_np as mnp
    n, train_adj, full_adj, _, _, _, labels, tr, va, te = synthetic.reddit_like(with_features=False)
    rng = np.random.RandomState(0)
    feats = rng.standard_normal((n, 602)).astype(np.float32)
    nbr = train_adj.dot(feats).astype(np.float32)
    fl = mnp.make_flags(normalization='graphsage', weight_decay=0.0, dropout=0.2, layer_norm=True, hidden1=128,
                        num_fc_layers=2, cv=True, cvd=True, degree=1, preprocess=True)
    case = dict(cfg=dict(model='vr', n=n, classes=41, batch=512), flags=fl, adj=train_adj, feats=feats, nbr=nbr,
                labels=labels, train=tr.astype(np.int32)[:4096], L_sched=1, ph=mc.placeholders(1, 41))
    _ffi.tune('step_fuse', 0)
    try:
        a, la = _run(case, False, 3, True)
    finally:
        _ffi.tune('step_fuse', 127)
    b, lb = _run(case, True, 3, True)
    assert all(p is not None for p in b._programs.values())
    assert torch.equal(a.theta, b.theta) and torch.equal(a.adam_m, b.adam_m) and torch.equal(a.adam_v, b.adam_v)
    for ha, hb in zip(a.history, b.history):
        assert torch.equal(ha[0], hb[0])
    for (l1, a1), (l2, a2) in zip(la, lb):
        assert torch.equal(l1, l2) and torch.equal(a1, a2)


SHAPES = {   # changes to the mid-size Reddit recipe (hidden1 64, f 96 -> 192 inputs, 41 classes, two pre-processing layers)
    'hidden128': dict(hidden1=128),                     # the widest layer the row passes fold
    'hidden256': dict(hidden1=256),                     # ... and one they must leave to the MFMA launches
    'hidden100': dict(hidden1=100),                     # 100 % 32 != 0: a partial K-step in every folded product
    'hidden32': dict(hidden1=32),                       # the narrowest layer whose backward weights go by swizzled direct loads
    'hidden30': dict(hidden1=30),                       # 30 % 4 != 0: no vector staging of the weights
    'one_fc': dict(num_fc_layers=1),                    # nothing to chain behind the upper layer's backward pass
    'three_fc': dict(num_fc_layers=3),                  # three parameter layers above one another
    'no_layer_norm': dict(layer_norm=False),            # ReLU only: no parameter partials, no split-K reduce pass to ride in
    'no_dropout': dict(dropout=0.0),
    'two_classes': dict(classes=2),
    'classes64': dict(classes=64),                      # the widest output layer the loss kernel takes as its head
    'classes70': dict(classes=70),                      # ... and one it does not
    'features50': dict(f=50),                           # 100 inputs: the first layer is not cut over K
    'batch7': dict(batch=7),                            # fewer rows than one workgroup's waves in the last layers
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_folded_products_on_other_shapes(shape):
    """Every folding decision of sgcn_step_run is a guard on widths (hidden size <= 128, classes <= 64, multiples of 4 for
    the vector staging, a layer cut over K or not, LayerNorm or not): the recipe on shapes on both sides of each guard, the
    step program with everything folded against the eager path with nothing folded -- the same bits."""
    from stochastic_gcn_amd import _ffi
    import copy
    cfg = copy.deepcopy(mc.REDDIT_MID)
    cfg.update(n=6000)
    for k, v in SHAPES[shape].items():
        if k in cfg:
            cfg[k] = v
        else:
            cfg['flags'][k] = v
    case = mc.build_case(cfg)
    _ffi.tune('step_fuse', 0)
    try:
        a, la = _run(case, False, 4, False)
    finally:
        _ffi.tune('step_fuse', 127)
    b, lb = _run(case, True, 4, False)
    if shape == 'hidden256':     # wider than the LayerNorm epilogue's 128 columns: no program, the per-layer path (whose
        assert 'wide' in b._program_note and not any(b._programs.values())     # backward pass still folds what fits)
    else:
        assert all(p is not None for p in b._programs.values())
    assert torch.equal(a.theta, b.theta) and torch.equal(a.adam_m, b.adam_m) and torch.equal(a.adam_v, b.adam_v)
    for ha, hb in zip(a.history, b.history):
        assert torch.equal(ha[0], hb[0])
    for (l1, a1), (l2, a2) in zip(la, lb):
        assert torch.equal(l1, l2) and torch.equal(a1, a2)


@pytest.mark.parametrize("fuse", [0, 127])
def test_multitask_model_sigmoid_loss_program_equals_eager_and_oracle(fuse):
    """The ppi form (gcn/models.py:77-79,86-90: multi-hot labels, sigmoid cross-entropy over all n x c elements): the loss
    kernel's other flavour, with the layers around it folded into its row pass or not -- bit-identical to the eager path,
    and the eager path's step against the oracle."""
    from stochastic_gcn_amd import _ffi
    from oracle import model_np as mnp, oracle_np as onp
    case = mc.build_case(mc.REDDIT_MID)
    rng = np.random.RandomState(5)
    case['labels'] = (rng.rand(*case['labels'].shape) < 0.2).astype(np.float32)        # multi-hot
    _ffi.tune('step_fuse', 0)
    try:
        a, la = _run(case, False, 3, False, multitask=True)
    finally:
        _ffi.tune('step_fuse', 127)
    _ffi.tune('step_fuse', fuse)
    try:
        b, lb = _run(case, True, 3, False, multitask=True)
    finally:
        _ffi.tune('step_fuse', 127)
    assert all(p is not None for p in b._programs.values())
    assert torch.equal(a.theta, b.theta) and torch.equal(a.adam_m, b.adam_m)
    for (l1, a1), (l2, a2) in zip(la, lb):
        assert torch.equal(l1, l2) and torch.equal(a1, a2)
    if fuse:
        return
    # the oracle on the same batches and masks
    fl, c, ph = case['flags'], case['cfg'], case['ph']
    om = mc.make_oracle_model(case, seed=3)
    om.multitask = True
    sch = mc.make_scheduler(case, 1)
    for step in range(3):
        feed = sch.minibatch(c['batch'])
        masks = mnp.HashMasks(a.dropout_seed, step, 1.0 - fl['dropout'])
        o_loss, o_acc, _, _, _ = om.run_one_step(feed, ph, fl['dropout'], masks)
        assert abs(float(la[step][0]) - float(o_loss)) <= 1e-4 * max(1.0, abs(float(o_loss))), (step, float(la[step][0]), float(o_loss))
        assert abs(float(la[step][1]) - float(o_acc)) <= 1e-5
    dp = a.get_params()
    for k, v in om.params.items():
        assert onp.rel_err(dp[k], v) < 5e-4, k
