"""End-to-end training on the GPU through the reference-shaped driver (train.Trainer): on a
graph whose labels come from a linear teacher on the 1-hop aggregate, Exact (PlainGCN,
degree >= max degree), NS+PP and CVD+PP must all learn; CVD+PP with degree 1 must end close to
Exact -- the paper's claim the reference README states (README.md:44) -- and the epoch log line
must keep the reference's token layout (scripts/analyze-time.py:39-54)."""
import io
import contextlib
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data():
    from stochastic_gcn_amd import synthetic
    return synthetic.reddit_like(n=6000, m=60000, f=32, classes=6, splits=(3600, 800, 1600), seed=5,
                                 with_features=True, planted=True)


def _train(flags, epochs):
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import Trainer
    FLAGS.reset()
    base = dict(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.1, layer_norm=True,
                hidden1=64, num_fc_layers=1, batch_size=256, test_batch_size=512, learning_rate=0.01, seed=1,
                prefetch=2)
    base.update(flags)
    FLAGS.update(**base)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        tr = Trainer(data=_data(), verbose=False)
        accs = []
        for _ in range(epochs):
            tr.train_epoch()
            accs.append(tr.evaluate(tr.val_d)[1])
    return tr, accs


def test_exact_ns_and_cvd_all_learn_and_cvd_matches_exact():
    _, exact = _train(dict(cv=False, degree=10000, test_degree=10000), 15)
    _, cvd = _train(dict(cv=True, cvd=True, test_cv=True, degree=1, test_degree=1), 15)
    _, ns = _train(dict(cv=False, degree=1, test_degree=10000), 15)
    print("val acc  exact %.3f  cvd+pp(d=1) %.3f  ns+pp(d=1) %.3f  (chance 0.167)" % (exact[-1], cvd[-1], ns[-1]))
    assert exact[-1] > 0.55 and exact[-1] > exact[0], exact
    assert cvd[-1] > 0.55 and cvd[-1] > cvd[0], cvd
    assert abs(cvd[-1] - exact[-1]) < 0.08, (cvd[-1], exact[-1])
    assert ns[-1] > 0.35


def test_driver_log_lines_and_counters():
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import Trainer
    FLAGS.reset()
    FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.2, layer_norm=True,
                 hidden1=32, num_fc_layers=2, batch_size=128, test_batch_size=256, cv=True, cvd=True,
                 test_cv=True, degree=1, test_degree=1, epochs=0, early_stopping=30, prefetch=0)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        tr = Trainer(data=_data())
        tr.SGDTrain()
        tr.Test()
    out = buf.getvalue()
    ep = [l for l in out.splitlines() if l.startswith("Epoch:")]
    assert len(ep) == 2       # the reference's exit is `epoch > FLAGS.epochs` (gcn/train.py:234): epochs + 2 epochs
    tok = ep[0].split()
    # token positions consumed by scripts/analyze-time.py:40-54 / plot-convergence.py:78-86
    assert tok[0] == "Epoch:" and tok[2] == "train_loss=" and tok[4] == "train_acc=" and tok[6] == "val_loss="
    assert tok[8] == "val_acc=" and "time=" in tok and "ttime=" in tok and "(sch" in tok and "data" in tok
    assert re.search(r"TF time = .*, g time = .*, G GFLOPS = .*, NN GFLOPS = .*, field sizes = ", out)
    assert re.search(r"Test set results: cost= \d+\.\d{5} accuracy= \d+\.\d{5} mi F1=", out)
    m = tr.train_model
    assert m.amt_data == 3600 and m.adj_sizes[0] == 3600          # degree 1: one sampled edge per train id
    assert m.fadj_sizes[0] > m.adj_sizes[0] and m.field_sizes[1] == 3600
    assert m.g_ops > 0 and m.nn_ops > 0


def test_training_is_bit_reproducible_and_prefetch_modes_agree():
    """Same seed -> bit-identical weights and history: the sampler is a deterministic stream, the
    dropout masks are a hash of (seed, layer, step, element), and no kernel uses float atomics
    (split rows, split-K and LayerNorm-parameter partials are all added in a fixed order).  The
    C++ prefetch thread, the Python prefetch thread and the synchronous loop are the same
    computation."""
    import torch
    runs = []
    for extra in (dict(prefetch=2, native_prefetch=True), dict(prefetch=2, native_prefetch=True),
                  dict(prefetch=2, native_prefetch=False), dict(prefetch=0)):
        tr, _ = _train(dict(cv=True, cvd=True, test_cv=True, degree=1, test_degree=1, **extra), 2)
        runs.append((tr.train_model.theta.clone(), tr.train_model.history[0][0].clone()))
    for theta, hist in runs[1:]:
        assert torch.equal(theta, runs[0][0])
        assert torch.equal(hist, runs[0][1])


def test_control_variate_predictions_scatter_less_than_neighbour_sampling():
    """What the control variate is for (the property the reference's --gradvar study measured, gcn/train.py:241-276;
    the study itself is out of scope): on one fixed batch, with the history warmed up by training and dropout off,
    ``get_pred_and_grad`` under the degree-2 training sampler scatters far less around the every-neighbour answer
    with the control variate than plain neighbour sampling does -- and the every-neighbour answer does not scatter."""
    from stochastic_gcn_amd.flags import FLAGS
    spread = {}
    for name, flags in (("ns", dict(cv=False, degree=2, test_degree=10000)),
                        ("cv", dict(cv=True, cvd=False, test_cv=False, degree=2, test_degree=10000))):
        tr, _ = _train(dict(gradvar=True, dropout=0.0, **flags), 20)
        ids = np.ascontiguousarray(tr.train_d[:FLAGS.batch_size], dtype=np.int32)

        def draws(sch, model, k):
            out = []
            for _ in range(k):
                feed = sch.batch(ids)
                feed[tr.placeholders['dropout']] = 0.0
                pred, _grad = model.get_pred_and_grad(tr.sess, feed)
                out.append(np.asarray(pred[0] if isinstance(pred, (list, tuple)) else pred, np.float64))
            return np.stack(out)
        exact = draws(tr.eval_sch, tr.test_model, 3)
        assert np.abs(exact - exact[0]).max() <= 1e-5 * np.abs(exact[0]).mean()     # every neighbour: deterministic
        part = draws(tr.train_sch, tr.train_model, 60)
        unit = np.abs(exact[0]).mean()
        spread[name] = (part.std(axis=0).mean() / unit, np.abs(part.mean(axis=0) - exact[0]).mean() / unit)
    print(spread)
    assert all(np.isfinite(v) for pair in spread.values() for v in pair)
    assert spread["cv"][0] < 0.5 * spread["ns"][0]
    assert spread["cv"][1] < spread["ns"][1]


def test_trainer_pp_products_run_the_column_sweep_and_match_scipy(tmp_path, monkeypatch):
    """The PP products of the training driver (gcn/utils.py:321-322), asked for often enough that a plan pays
    (--pp_products 1000), go through sgcn_spmm_cs_f32 -- the kernel bench.py times -- with the host plan cached beside the
    dataset; the result equals SciPy's (the reference's own library for this product) within 1e-4.  (Run once, as in the
    reference: test_pp_products_run_once_take_the_rows_kernel.)"""
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import Trainer
    from oracle import oracle_np as onp
    monkeypatch.setenv("SGCN_PLAN_CACHE_DIR", str(tmp_path))
    data = _data()
    n, train_adj, full_adj, feats = data[0], data[1], data[2], data[3]
    stats = []
    for _ in range(2):
        FLAGS.reset()
        FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.1, layer_norm=True,
                     hidden1=32, num_fc_layers=1, batch_size=256, test_batch_size=512, cv=True, cvd=True,
                     test_cv=True, degree=1, test_degree=1, pp_products=1000)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            tr = Trainer(data=data, verbose=False)
        stats.append(tr.pp_stats)
        f = feats.shape[1]
        for model, adj in ((tr.train_model, train_adj), (tr.test_model, full_adj)):
            got = model.features_dev[:, f:].cpu().numpy()
            want = adj.dot(feats).astype(np.float32)
            assert onp.rel_err(got, want) <= 1e-4
            assert np.array_equal(model.features_dev[:, :f].cpu().numpy(), feats)
    assert len(stats[0]) == 2 and all("cs_spmm" in s["kernel"] for s in stats[0] + stats[1])
    from stochastic_gcn_amd import ops
    assert ops.ColumnSweepCSR.choose_g(feats.shape[1]) == 2                      # f = 32: one 128-column pass, two lane groups
    assert all("cs_spmm16g2k" in s["kernel"] for s in stats[0] + stats[1])      # ... built, cached and re-loaded as such
    assert [s["plan_from_cache"] for s in stats[0]] == [False, False]
    assert [s["plan_from_cache"] for s in stats[1]] == [True, True]          # second run: plans (and paces) from disk
    assert [s["pace"] for s in stats[0]] == [s["pace"] for s in stats[1]]
    assert len(list(tmp_path.glob("*.csplan.*.npz"))) == 2


@pytest.mark.parametrize("weight_decay", [0.0, 5e-4])
def test_evaluation_through_the_step_program_equals_the_eager_path(weight_decay):
    """Evaluation (gcn/train.py:133-160: forward + loss + prediction + the TEST model's own history scatter, with
    --test_cv warm-up sweeps reading what the previous sweep wrote) as compiled step programs -- one foreign call per
    batch -- against the eager per-layer path: loss, accuracy, both F1 scores and the test history, bit for bit.  With the
    default weight decay (5e-4) the reported cost includes the L2 term of the first parametrised layer on BOTH paths
    (gcn/models.py:75; the eager path once snapshotted its result vector in front of sgcn_l2_penalty_f32: ADVICE r4)."""
    import torch
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import Trainer
    res = {}
    for native in (False, True):
        FLAGS.reset()
        FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=weight_decay, dropout=0.1, layer_norm=True,
                     hidden1=64, num_fc_layers=2, batch_size=256, test_batch_size=512, learning_rate=0.01, seed=1,
                     prefetch=2, cv=True, cvd=True, test_cv=True, degree=1, test_degree=1, native_step=native)
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Trainer(data=_data(), verbose=False)
            tr.train_epoch()
            sweeps = [tr.evaluate(tr.val_d)[:4] for _ in range(2)]
            sweeps.append(tr.evaluate(tr.test_d)[:4])
        torch.cuda.synchronize()
        progs = getattr(tr.test_model, '_programs', {})
        res[native] = (sweeps, tr.test_model.history[0][0].clone(), bool(progs) and all(p is not None for p in progs.values()),
                       getattr(tr.test_model, '_program_note', None))
    assert res[True][2], res[True][3]                 # the test model really ran as a program ...
    assert not res[False][2]
    assert res[True][0] == res[False][0], (res[True][0], res[False][0])      # ... and gave the same numbers
    assert torch.equal(res[True][1], res[False][1]) and float(res[True][1].abs().sum()) > 0
    assert res[True][0][0] != res[True][0][1]         # the second sweep read the history the first one wrote
    if weight_decay:
        # the cost carries the L2 term of the first parametrised layer (the last trainer is the program one; its eager
        # twin reported the same numbers above): one more eager batch read through BOTH result forms
        m = tr.test_model
        w = m.theta[m._wd_range[0]:m._wd_range[1]]
        l2 = float(weight_decay * 0.5 * (w.double() ** 2).sum())
        assert l2 > 0 and all(sw[0] > l2 for sw in res[False][0])
        FLAGS.native_step = False
        batch = tr.eval_sch.batch_packed(tr.val_d[:256], FLAGS.plan_t, tr.eval_slots[0])
        m.eval_light, m.eval_sink = True, None
        try:
            los, acc, _ = m.run_one_step(None, batch, sync=True)
        finally:
            m.eval_light = False
        vec = m.__dict__.pop('eval_vec')
        assert float(vec[2]) == float(los) and float(vec[3]) == float(acc)


def test_f1_scores_from_the_loss_kernels_class_indices_are_sklearns():
    """An evaluation sweep brings argmax(pred) + 4096 * argmax(labels) per row to the host (third plane of the loss
    kernel's row scratch) instead of the prediction and label matrices: the indices against np.argmax of what
    run_one_step returns, and Trainer.evaluate's F1 pair against gcn/utils.py:521-529's sklearn call on the matrices --
    on both step paths."""
    import numpy as np
    import torch
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import Trainer
    from stochastic_gcn_amd.utils import calc_f1
    for native in (False, True):
        FLAGS.reset()
        FLAGS.update(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.1, layer_norm=True,
                     hidden1=64, num_fc_layers=2, batch_size=256, test_batch_size=512, learning_rate=0.01, seed=1,
                     prefetch=2, cv=True, cvd=True, test_cv=True, degree=1, test_degree=1, native_step=native)
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Trainer(data=_data(), verbose=False)
            tr.train_epoch()
            preds, labs, clss = [], [], []
            orig = tr.test_model.run_one_step

            def spy(sess, batch, sync=True):
                tr.test_model.eval_light = False          # (the full outputs: evaluate then takes the class indices per batch)
                out = orig(sess, batch, sync=sync)
                preds.append(out[2].clone()); labs.append(tr.test_model.cur.labels.clone())
                clss.append(tr.test_model.eval_classes.clone())
                return out
            tr.test_model.run_one_step = spy
            _, _, micro, macro, _ = tr.evaluate(tr.val_d)
        pred, lab = torch.cat(preds).cpu().numpy(), torch.cat(labs).cpu().numpy()
        v = torch.cat(clss).cpu().numpy().astype(np.int64)
        assert np.array_equal(v % 4096, pred.argmax(1)) and np.array_equal(v // 4096, lab.argmax(1))
        want = calc_f1(pred, lab, False)
        assert abs(micro - want[0]) < 1e-12 and abs(macro - want[1]) < 1e-12 and 0 < micro < 1


def test_multilabel_evaluation_sums_on_the_host_and_matches_sklearn():
    """--dataset ppi (multi-label: sigmoid cross-entropy, gcn/models.py:77-94): Trainer.evaluate's prediction / label form
    -- every batch's loss, accuracy, predictions and labels go to pinned host memory as they are produced, the sums are
    taken on the host after ONE synchronisation (round 6: no torch.stack / cat / sum on this path) -- gives sklearn's
    micro / macro F1 of the thresholded predictions, the row-weighted mean loss, and the same numbers program or eager."""
    import torch
    from sklearn.metrics import f1_score
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import Trainer
    res = {}
    for native in (True, False):
        data = list(_data())               # (fresh per trainer: the sampler shuffles the training ids in place)
        data[6] = (np.random.RandomState(0).rand(*data[6].shape) < 0.3).astype(np.float32)      # several labels per vertex
        FLAGS.reset()
        FLAGS.update(dataset='ppi', normalization='graphsage', weight_decay=0.0, dropout=0.1, layer_norm=True, hidden1=64,
                     num_fc_layers=1, batch_size=256, test_batch_size=300, learning_rate=0.01, seed=1, prefetch=2, cv=True, cvd=True,
                     test_cv=True, degree=1, test_degree=1, native_step=native)
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Trainer(data=tuple(data), verbose=False)
            assert tr.multitask
            tr.train_epoch()
            preds, labs, stats = [], [], []
            orig = tr.test_model.run_one_step

            def spy(sess, batch, sync=True):
                out = orig(sess, batch, sync=sync)
                preds.append(out[2].clone()); labs.append(tr.test_model.cur.labels.clone())
                stats.append((float(out[0]), float(out[1]), int(out[2].shape[0])))
                return out
            tr.test_model.run_one_step = spy
            cost, acc, micro, macro, _ = tr.evaluate(tr.val_d)
        pred, lab = torch.cat(preds).cpu().numpy(), torch.cat(labs).cpu().numpy()
        n = sum(r for _, _, r in stats)
        assert n == len(tr.val_d) and len(stats) == -(-n // 300)
        assert abs(cost - sum(c * r for c, _, r in stats) / n) <= 1e-5 * max(1.0, abs(cost))
        assert abs(acc - sum(a * r for _, a, r in stats) / n) <= 1e-6
        hard = (pred > 0.5).astype(np.int64)
        assert abs(micro - f1_score(lab, hard, average="micro")) < 1e-9 and abs(macro - f1_score(lab, hard, average="macro")) < 1e-9
        assert 0 < micro < 1
        res[native] = (cost, acc, micro, macro)
    assert np.allclose(res[True], res[False], rtol=1e-5)


def test_no_library_gemm_on_the_product_path(monkeypatch):
    """the size-keyed rocBLAS path (torch.mm above 512 M multiply-adds: Exact mode, large
    evaluation batches) is gone.  With every torch matmul entry point booby-trapped, the Reddit recipe (CVD+PP,
    program and eager) and an Exact configuration whose dense layers see ALL 6,000 vertices at once -- 6,000 x 64 x 96
    and the like, and an evaluation batch of 6,000 -- still train and evaluate: every dense product ran on
    sgcn_gemm.hip."""
    import torch
    from stochastic_gcn_amd.flags import FLAGS
    from stochastic_gcn_amd.train import Trainer

    def trap(*a, **k):
        raise AssertionError("a library GEMM (torch.mm / addmm / matmul) was reached on the product path")
    for name in ("mm", "addmm", "matmul", "bmm"):
        monkeypatch.setattr(torch, name, trap)
    monkeypatch.setattr(torch.Tensor, "addmm_", trap)
    monkeypatch.setattr(torch.Tensor, "mm", trap)
    monkeypatch.setattr(torch.Tensor, "matmul", trap)
    monkeypatch.setattr(torch.Tensor, "__matmul__", trap)
    base = dict(dataset='s-reddit', normalization='graphsage', weight_decay=0.0, dropout=0.1, layer_norm=True,
                hidden1=64, num_fc_layers=2, learning_rate=0.01, seed=1, prefetch=2)
    for flags in (dict(cv=True, cvd=True, test_cv=True, degree=1, test_degree=1, batch_size=256, test_batch_size=512),
                  dict(cv=True, cvd=True, test_cv=True, degree=1, test_degree=1, batch_size=256, test_batch_size=512, native_step=False),
                  dict(cv=False, degree=10000, test_degree=10000, batch_size=6000, test_batch_size=6000, preprocess=False,
                       num_fc_layers=1)):
        FLAGS.reset()
        FLAGS.update(**dict(base, **flags))
        with contextlib.redirect_stdout(io.StringIO()):
            tr = Trainer(data=_data(), verbose=False)
            tr.train_epoch()
            cost, acc = tr.evaluate(tr.val_d)[:2]
        assert np.isfinite(cost) and 0.0 <= acc <= 1.0


def test_det_dropout_model_trains_through_the_driver():
    """--det_dropout (moment propagation instead of sampled dropout, gcn/train.py:55) with the CV estimator's two
    histories per layer: the driver trains it (eager layers, packed minibatches) and it learns the planted labels."""
    tr, accs = _train(dict(cv=True, cvd=False, det_dropout=True, test_cv=True, degree=2, test_degree=2, dropout=0.2,
                           num_layers=3), 8)
    assert all(len(h) == 2 for h in tr.train_model.history)
    assert all(float(h.abs().max()) > 0 for hs in tr.train_model.history for h in hs)
    assert np.isfinite(accs).all() and accs[-1] > 0.45 and accs[-1] > accs[0], accs


def test_pp_products_pick_the_lds_sweep_for_a_graph_with_communities():
    """train.pp_products on a graph whose nonzeros sit inside communities (p_in 0.95) runs the LDS-staged sweep + its
    residual and agrees with SciPy (the reference's own library for this product, gcn/utils.py:321-322); on a graph
    without structure it keeps the column sweep."""
    import torch
    from stochastic_gcn_amd import synthetic, train
    dev = torch.device("cuda:0")
    data = synthetic.reddit_sbm(n=60000, m=3000000, classes=12, splits=(40000, 8000, 12000), p_in=0.95, seed=4)
    a = data[2]
    rng = np.random.RandomState(0)
    X = rng.standard_normal((a.shape[0], 130)).astype(np.float32)
    stats = []
    tf, ff = train.pp_products(data[1], a, X, dev, stats=stats, products=5000)
    assert "lds_spmm_kernel" in stats[1]["kernel"], stats
    for got, m in ((tf, data[1]), (ff, a)):
        ref = m.astype(np.float64).dot(X.astype(np.float64))
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    flat = synthetic.reddit_like(n=60000, m=3000000, f=8, classes=5, splits=(40000, 8000, 12000), seed=4, with_features=False)
    stats = []
    train.pp_products(flat[1], flat[2], X, dev, stats=stats, products=5000)
    assert all("cs_spmm" in s_["kernel"] for s_ in stats), stats


def test_pp_products_run_once_take_the_rows_kernel(tmp_path, monkeypatch):
    """The reference computes each PP product ONCE (gcn/utils.py:321-322).  For one product no plan pays (bench.py's
    setup.products_to_break_even_vs_rows_kernel: ~83 on S-Reddit): train.pp_products then runs the row-gather kernel, writes
    no plan cache, and agrees with SciPy; the choice flips to the column sweep where the arithmetic of
    train.static_kernel_for says so."""
    import torch
    from stochastic_gcn_amd import synthetic, train
    monkeypatch.setenv("SGCN_PLAN_CACHE_DIR", str(tmp_path))
    dev = torch.device("cuda:0")
    flat = synthetic.reddit_like(n=60000, m=3000000, f=8, classes=5, splits=(40000, 8000, 12000), seed=4, with_features=False)
    X = np.random.RandomState(0).standard_normal((flat[2].shape[0], 130)).astype(np.float32)
    stats = []
    tf, ff = train.pp_products(flat[1], flat[2], X, dev, cache=(str(tmp_path / "a.npz"), str(tmp_path / "b.npz")), stats=stats)
    assert all("spmm_seg_kernel" in s_["kernel"] for s_ in stats), stats
    assert not list(tmp_path.glob("*.npz"))
    for got, m in ((tf, flat[1]), (ff, flat[2])):
        ref = m.astype(np.float64).dot(X.astype(np.float64))
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    assert train.static_kernel_for(23173306, 602, 1) == 'rows' and train.static_kernel_for(23173306, 602, 40) == 'rows'
    assert train.static_kernel_for(23173306, 602, 120) == 'cs'
    n = [k for k in range(1, 400) if train.static_kernel_for(23173306, 602, k) == 'cs'][0]
    assert 70 <= n <= 100, n            # (measured: 83, profiles/r60_bench_setup.json)
