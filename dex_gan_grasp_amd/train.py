"""Thin train-loop driver: the loop of the reference's train.py (:231-310) over batch iterables, on the drop-in wrapper.

The reference trains with one epoch loop per network — every batch of the generator data set through
`update_dexgangrasp(data, is_train_gen = (i % gen_train_freq == 0))` (:243-252), then every batch of the evaluator data set
through `update_dexevaluator(data)` (:267-277) —, saves both every `save_freq` epochs (:289-295) and evaluates right after
(:302-306).  `train()` reproduces that order ("reference" mode) and offers the order the MI355X path is fastest in
("interleaved" mode): generator batch i and evaluator batch i go through `DexGanGrasp.update_joint` as ONE step whose
two updates run side by side, with a one-batch look-ahead on the generator batches (the next batch's discriminator
real-batch pass is enqueued one step early).  The two updates touch disjoint networks, optimisers and batches, so both
modes end an epoch with the same parameters, bit for bit (tests/test_gpu_train_loop.py).

What is NOT here: the reference's h5 data sets (not shipped; `data.synthetic_data_set.SyntheticGraspBatches` stands in),
its tensorboard Writer and its argument parser — `on_train_loss` / `on_epoch_end` / `eval_fn` are plain callables.

    python -m dex_gan_grasp_amd.train --epochs 2 --batches 8 --batch-size 256 --mode interleaved
"""
from __future__ import annotations

import argparse
import time


def _resident(batch):
    """A batch dict whose tensors the look-ahead may hand to the device path one step early."""
    return batch if all(hasattr(batch.get(k), "is_cuda") for k in ("bps_object", "rot_matrix", "transl", "joint_conf")) else None


def train(model, cfg, gen_batches, eva_batches, mode="reference", start_epoch=None, eval_fn=None, on_train_loss=None,
          on_epoch_end=None, save=True):
    """model: models.dexgangrasp.DexGanGrasp (is_train).  gen_batches / eva_batches: re-iterable batch sources (a
    DataLoader, a SyntheticGraspBatches, a list of dicts) or None when that network is not trained.  cfg keys as in
    train.py: num_epochs, gen_train_freq, save_freq, print_freq, batch_size, continue_train / load_epoch.
    eval_fn(model, epoch) -> dict is called where train.py calls run_eval_gan (:302-306); on_train_loss(epoch, epoch_iter,
    loss_dict) where it prints / plots the train loss (:255-259, :280-284).  Returns the number of samples seen."""
    if mode not in ("reference", "interleaved"):
        raise ValueError("mode must be 'reference' or 'interleaved'")
    train_gen = bool(getattr(model, "train_dexgenerator", False)) and gen_batches is not None
    train_eva = bool(getattr(model, "train_dexevaluator", False)) and eva_batches is not None
    if start_epoch is None:
        if cfg.get("continue_train"):   # train.py:219-225
            if train_eva:
                model.load_dexevaluator(cfg["load_epoch"])
            if train_gen:
                model.load_dexgenerator(cfg["load_epoch"])
            start_epoch = cfg["load_epoch"] + 1
        else:
            start_epoch = 1
    freq = max(1, int(cfg.get("gen_train_freq", 1)))
    bs = int(cfg.get("batch_size", 0))
    pf = int(cfg.get("print_freq", 0))
    total_steps = 0
    t_start = time.time()

    def log(epoch, epoch_iter, loss_dict):
        if on_train_loss is not None and pf > 0 and total_steps % pf == 0:
            on_train_loss(epoch, epoch_iter, loss_dict)

    for epoch in range(start_epoch, int(cfg["num_epochs"]) + 1):
        if mode == "reference" or not (train_gen and train_eva):
            if train_gen:                                   # === Generator === (:233-262)
                epoch_iter = 0
                for i, data in enumerate(gen_batches):
                    total_steps += bs
                    epoch_iter += bs
                    loss_dict = model.update_dexgangrasp(data, is_train_gen=(i % freq == 0))
                    log(epoch, epoch_iter, loss_dict)
            if train_eva:                                   # === Evaluator === (:264-286)
                epoch_iter = 0
                for i, data in enumerate(eva_batches):
                    total_steps += bs
                    epoch_iter += bs
                    loss_dict = model.update_dexevaluator(data)
                    log(epoch, epoch_iter, loss_dict)
        else:
            # generator batch i beside evaluator batch i; whichever data set is longer finishes on its own
            git, eit = iter(gen_batches), iter(eva_batches)
            cur = next(git, None)
            i = 0
            epoch_iter = 0
            while cur is not None:
                nxt = next(git, None)
                ev = next(eit, None)
                total_steps += bs
                epoch_iter += bs
                if ev is not None:
                    ahead = _resident(nxt) if nxt is not None else None   # (used only when this step trains the generator)
                    loss_dict, loss_eva = model.update_joint(cur, ev, is_train_gen=(i % freq == 0), next_real_data=ahead)
                    total_steps += bs
                    log(epoch, epoch_iter, loss_eva)
                else:
                    loss_dict = model.update_dexgangrasp(cur, is_train_gen=(i % freq == 0))
                log(epoch, epoch_iter, loss_dict)
                cur = nxt
                i += 1
            for ev in eit:
                total_steps += bs
                log(epoch, epoch_iter, model.update_dexevaluator(ev))
        if save and epoch % int(cfg["save_freq"]) == 0:      # (:289-295)
            print('Saving the model after epoch %d, iters %d' % (epoch, total_steps))
            if train_gen:
                model.save_dexgenerator(str(epoch), epoch)
            if train_eva:
                model.save_dexevaluator(str(epoch), epoch)
        print('End of epoch %d / %d \t Time taken: %.3f min' % (epoch, cfg["num_epochs"], (time.time() - t_start) / 60))
        if eval_fn is not None and epoch % int(cfg["save_freq"]) == 0:   # (:302-306)
            eval_loss_dict = eval_fn(model, epoch)
            if on_epoch_end is not None:
                on_epoch_end(epoch, eval_loss_dict)
    return total_steps


def main(argv=None):
    import numpy as np
    import torch
    from . import synth
    from .data.synthetic_data_set import SyntheticGraspBatches
    from .models.dexgangrasp import DexGanGrasp
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batches", type=int, default=8, help="batches per epoch and data set")
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--points", type=int, default=2048, help="points per synthetic object cloud")
    ap.add_argument("--mode", choices=["reference", "interleaved"], default="interleaved")
    ap.add_argument("--save-dir", default=None, help="checkpoint directory (default: none written)")
    ap.add_argument("--prefetch", action="store_true", help="loaders one batch ahead: BPS encode on a side stream beside the step")
    a = ap.parse_args(argv)
    cfg = dict(synth.DEFAULT_CFG)
    cfg.update({"num_epochs": a.epochs, "batch_size": a.batch_size, "train_ffhevaluator": True, "train_ffhgenerator": True,
                "save_freq": 1, "print_freq": a.batch_size * max(1, a.batches // 2), "gen_train_freq": 1})
    if a.save_dir:
        cfg["save_dir"] = a.save_dir
    model = DexGanGrasp(cfg)
    g = torch.Generator().manual_seed(0)
    basis = (torch.rand((4096, 3), generator=g) * 0.2 - 0.1).numpy().astype(np.float32)
    gen = SyntheticGraspBatches(cfg, a.batch_size, a.batches, basis, n_points=a.points, seed=1, device_tensors=True,
                                prefetch=a.prefetch)
    eva = SyntheticGraspBatches(cfg, a.batch_size, a.batches, basis, n_points=a.points, seed=2, with_label=True,
                                device_tensors=True, prefetch=a.prefetch)
    t0 = time.time()
    n = train(model, cfg, gen, eva, mode=a.mode, save=bool(a.save_dir),
              on_train_loss=lambda ep, it, ld: print(ep, it, {k: round(float(v), 5) for k, v in ld.items()}))
    torch.cuda.synchronize()
    print("%d samples in %.2f s (%s order, encode included)" % (n, time.time() - t0, a.mode))


if __name__ == "__main__":
    main()
