#!/usr/bin/env python3
"""Bisect a non-finite SimCLR training loss at BASELINE cfg 2 (bs 256, bf16, lr 0.06).

    python tools/nan_hunt.py --mode eager|graph --steps 150 [--lr-scale 0.1] [--fmt nhwc_bf16]
                             [--no-fused-stats] [--momentum 0.9] [--out gpurun_out/nan_x.jsonl]

Per step it logs the loss, the rep_std, the gradient / parameter arena norms and maxima, the count of
non-finite gradient entries, the largest BatchNorm 1/sigma and the smallest / largest BN weight.  At the
first non-finite loss or gradient (checked BEFORE the optimiser step, so the weights are still the
ones that produced it) the same batch is run again with forward hooks and every module's output is
checked: the first module with a non-finite output is reported, plus the parameters whose gradient
slots hold non-finite values.
"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="eager", choices=["eager", "graph"])
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--wafers", type=int, default=4096)
    ap.add_argument("--fmt", default="s2d_bf16")
    ap.add_argument("--lr-scale", type=float, default=1.0)
    ap.add_argument("--momentum", type=float, default=0.9)
    ap.add_argument("--no-fused-stats", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--every", type=int, default=1)
    ap.add_argument("--probe", action="store_true",
                    help="record max|.| of every module output and of its gradient into a static buffer (captured "
                         "into the graph too), and print them in execution order at the first non-finite step")
    ap.add_argument("--probe-max-bytes", type=int, default=1 << 40,
                    help="probe only tensors up to this size (<= 512 KiB keeps the probes' temporaries in the caching "
                         "allocator's small-block pool, so the large-block reuse pattern of the captured graph is unchanged)")
    ap.add_argument("--poison", action="store_true",
                    help="fill every torch.empty / empty_like allocation with NaN (floats) or 0xFF (integers): a kernel "
                         "that reads memory it (or its producer) never wrote then shows up deterministically")
    a = ap.parse_args()

    import numpy as np
    import torch

    from ssl_wafermap_amd import _lib, ops
    from ssl_wafermap_amd.data import WaferMapDataset
    from ssl_wafermap_amd.data.synthetic import synthetic_wafers
    from ssl_wafermap_amd.models import SimCLR
    from ssl_wafermap_amd.transforms import BaseViewTransform

    if a.poison:
        _empty, _empty_like = torch.empty, torch.empty_like

        def _fill(t):
            if t.is_cuda and t.numel():
                if t.is_floating_point():
                    t.fill_(float("nan"))
                elif t.dtype == torch.uint8:
                    t.fill_(0xFF)
                elif t.dtype in (torch.int32, torch.int64, torch.int16):
                    t.fill_(-1)
            return t

        torch.empty = lambda *x, **k: _fill(_empty(*x, **k))
        torch.empty_like = lambda *x, **k: _fill(_empty_like(*x, **k))

    if a.no_fused_stats:
        ops.stats_fusable = lambda rows, groups: False
        import ssl_wafermap_amd.models.resnet  # noqa: F401  (uses ops.stats_fusable through ops)

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B = a.batch
    wafers, labels = synthetic_wafers(a.wafers, seed=1234)
    ds = WaferMapDataset(wafers, labels, transform=BaseViewTransform(), device=dev)
    torch.manual_seed(0)
    model = SimCLR(None, 9, batch_size=B, max_epochs=150).to(dev).train()
    (opt,), _ = model.configure_optimizers()
    for g in opt.param_groups:
        g["lr"] *= a.lr_scale
        g["momentum"] = a.momentum
    rng = np.random.default_rng(0)
    arena = opt._arenas[0]
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    params = [p for _, p in model.named_parameters() if p.requires_grad]
    bns = [(n, m) for n, m in model.named_modules() if hasattr(m, "running_var")]

    probe_names, probe_buf = [], None
    if a.probe:
        probe_buf = torch.zeros(4096, dtype=torch.float32, device=dev)
        slots = {}

        def slot(name):
            if name not in slots:
                slots[name] = len(probe_names)
                probe_names.append(name)
            return slots[name]

        def probe_rec(name, t):
            # max propagates NaN; abs of inf stays inf
            if t.numel() * t.element_size() <= a.probe_max_bytes and t.dtype in (torch.float32, torch.bfloat16) \
                    and t.is_contiguous(memory_format=torch.channels_last if t.dim() == 4 else torch.contiguous_format):
                i = slot(name)
                _lib.check(_lib.load().wm_debug_absmax(t.data_ptr(), _lib.dtype_code(t), t.numel(),
                                                       probe_buf.data_ptr() + 4 * i, _lib.stream_ptr()), "wm_debug_absmax")

        def fwd_hook(name):
            def f(mod, inp, outp):
                ts = outp if isinstance(outp, (tuple, list)) else (outp,)
                for k, t in enumerate(ts):
                    if torch.is_tensor(t) and t.is_floating_point():
                        probe_rec(f"fwd {name}[{k}]", t)
                        if t.requires_grad:
                            t.register_hook(lambda g, nm=f"bwd d({name}[{k}])": probe_rec(nm, g))
            return f

        for n, m in model.named_modules():
            if n:
                m.register_forward_hook(fwd_hook(n))
        # inside the loss: inputs and output of the NT-Xent backward kernel, and of the normalisation backward
        from ssl_wafermap_amd import functional as Fh

        _nb = Fh.ntxent_backward

        def ntxent_backward_probed(zn, zall, lse_all, *rest):
            probe_rec("loss: ntxent_bwd in zn", zn)
            probe_rec("loss: ntxent_bwd in lse_all", lse_all)
            dzn = _nb(zn, zall, lse_all, *rest)
            probe_rec("loss: ntxent_bwd out dzn", dzn)
            return dzn

        Fh.ntxent_backward = ntxent_backward_probed
        _l2b = Fh._L2Normalize.backward

        def l2_backward_probed(ctx, dy):
            probe_rec("loss: l2norm_bwd in dy", dy if dy.is_contiguous() else dy.contiguous())
            r = _l2b(ctx, dy)
            probe_rec("loss: l2norm_bwd out dx", r[0])
            return r

        Fh._L2Normalize.backward = staticmethod(l2_backward_probed)

    def dump_probe():
        if probe_buf is None:
            return
        vals = probe_buf[: len(probe_names)].cpu().tolist()
        emit({"probe": [[n, v] for n, v in zip(probe_names, vals)]})

    graphed = None
    if a.mode == "graph":
        from ssl_wafermap_amd.graph import GraphedTrainStep

        for i in range(3):
            batch = ds.get_batch((np.arange(B) + i * B) % len(ds), rng, fmt=a.fmt)
            opt.zero_grad()
            model.training_step(batch, i).backward()
            opt.step()
        graphed = GraphedTrainStep(model, opt, ds, B, fmt=a.fmt).capture(np.arange(B), rng, None)

    out = open(a.out, "w") if a.out else None

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + "\n")
            out.flush()

    def diagnose(batch, step):
        bad_mods = []

        def hook(name):
            def f(mod, inp, outp):
                ts = outp if isinstance(outp, (tuple, list)) else (outp,)
                for t in ts:
                    if torch.is_tensor(t) and t.is_floating_point():
                        tf = t.float()
                        fin = bool(torch.isfinite(tf).all())
                        bad_mods.append((name, type(mod).__name__, fin, float(tf.abs().max()) if fin else float("nan")))
            return f

        hs = [m.register_forward_hook(hook(n)) for n, m in model.named_modules() if n]
        with torch.no_grad():
            loss = model.training_step(batch, step)
        for h in hs:
            h.remove()
        first_bad = next(((n, t) for n, t, fin, _ in bad_mods if not fin), None)
        big = sorted(((mx, n) for n, t, fin, mx in bad_mods if fin), reverse=True)[:8]
        bad_grads = [n for n, p in zip(names, params) if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        bad_params = [n for n, p in zip(names, params) if not bool(torch.isfinite(p).all())]
        emit({"diagnosis_step": step, "rerun_loss": float(loss), "first_nonfinite_module": first_bad,
              "largest_outputs": big, "nonfinite_grads": bad_grads[:40], "n_nonfinite_grads": len(bad_grads),
              "nonfinite_params": bad_params[:40]})

    for i in range(a.steps):
        idx = (np.arange(B) + i * B) % len(ds)
        if probe_buf is not None:
            probe_buf.zero_()
        if graphed is None:
            batch = ds.get_batch(idx, rng, fmt=a.fmt)
            opt.zero_grad()
            loss = model.training_step(batch, i)
            loss.backward()
        else:
            graphed._upload(graphed.tr.sample(ds.store, idx, rng))
            graphed.graph.replay()
            loss = graphed.loss
            batch = None
        lv = float(loss.detach())
        g = arena.grads
        nf = int((~torch.isfinite(g)).sum())
        rec = {"step": i, "loss": lv, "grad_norm": float(g.norm()), "grad_max": float(g.abs().max()),
               "grad_nonfinite": nf, "param_norm": float(arena.params.norm()), "param_max": float(arena.params.abs().max()),
               "mom_max": float(arena.momentum.abs().max())}
        if i % a.every == 0 or nf or lv != lv:
            rv = [(float(m.running_var.min()), n) for n, m in bns]
            rec["min_running_var"] = min(rv)
            rec["bn_w_absmax"] = max(float(m.weight.abs().max()) for _, m in bns)
            rs = model.logged.get("rep_std")
            rec["rep_std"] = float(rs) if rs is not None else None
            emit(rec)
        if nf or not np.isfinite(lv):
            dump_probe()
            if graphed is not None:
                if probe_buf is not None:
                    probe_buf.zero_()
                graphed.graph.replay()
                emit({"second_replay_same_inputs": True, "loss": float(graphed.loss.detach()),
                      "grad_nonfinite": int((~torch.isfinite(arena.grads)).sum())})
                dump_probe()
                # same decisions (still in the static parameter buffers), same weights, eagerly
                if probe_buf is not None:
                    probe_buf.zero_()
                opt.zero_grad()
                views = graphed.tr.launch(ds.store, graphed._last_params, B, a.fmt, params_dev=graphed.static)
                l2 = model.training_step((views, None), i)
                l2.backward()
                emit({"eager_rerun_same_inputs": True, "loss": float(l2.detach()),
                      "grad_nonfinite": int((~torch.isfinite(arena.grads)).sum())})
                dump_probe()
            if batch is None:
                batch = ds.get_batch(idx, np.random.default_rng(0), fmt=a.fmt)
            diagnose(batch, i)
            break
        opt.step()
    torch.cuda.synchronize()
    emit({"done": True, "mode": a.mode, "steps_run": i + 1})


if __name__ == "__main__":
    main()
