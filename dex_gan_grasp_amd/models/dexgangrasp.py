"""Drop-in mirror of DexGanGrasp/models/dexgangrasp.py (class DexGanGrasp, reference lines cited per
method): same constructor (cfg dict), method names, loss-dict keys, ValueError/NotImplementedError
behaviour and checkpoint layout, with the training steps and inference running in libdgg.so.

The training steps (update_dexgangrasp / update_dexevaluator) do not go through torch autograd:
they call the fused phases of the C-ABI (forward, losses, backward, Adam entirely on the GPU, no
host synchronisation); the returned loss dict holds 0-d device tensors like the reference's.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .. import _lib, ops, parallel
from ..engine import TrainEngine
from ..utils import utils
from ..utils.train_tools import EarlyStopping
from . import losses, networks
from .networks import DexEvaluator, DexGANGrasp


# DGG_API_STREAMS=0: run update_* on the caller's stream only (default: D's real-batch pass and the weight-gradient
# GEMMs of each update overlap on library side streams, joined before the call returns; bit-identical results)
_API_STREAMS = os.environ.get("DGG_API_STREAMS", "1") != "0"


def define_losses(loss_type_recon):
    """dexgangrasp.py:13-25 (the VAE-era KL term is dead code in the reference: not mirrored)."""
    kl_loss = None
    if loss_type_recon == 'transl_rot_6D_l2':
        reconstruction_loss = losses.transl_rot_6D_l2_loss
    else:
        raise Exception('Requested loss not available, choose [transl_rot_6D_l2]')
    return kl_loss, reconstruction_loss


def build_network(cfg, device, is_train):
    """dexgangrasp.py:28-37."""
    gen = DexGANGrasp(cfg)
    eva = DexEvaluator(cfg)
    gen.to(device)
    eva.to(device)
    if is_train:
        init_net(gen)
        init_net(eva)
    return gen, eva


# weight_init_type -> initialiser of a Linear / Conv weight (same names, distributions and gains as dexgangrasp.py:39-62)
_WEIGHT_INIT = {
    "normal": lambda w, gain: torch.nn.init.normal_(w, 0.0, gain),
    "xavier": lambda w, gain: torch.nn.init.xavier_normal_(w, gain=gain),
    "kaiming": lambda w, gain: torch.nn.init.kaiming_normal_(w, a=0, mode="fan_in"),
    "orthogonal": lambda w, gain: torch.nn.init.orthogonal_(w, gain=gain),
}


def init_net(net):
    """Initialise a freshly built network the way the reference does (dexgangrasp.py:39-62): every Linear / Conv weight
    by cfg["weight_init_type"] with cfg["init_gain"], every BatchNorm weight ~ N(1, gain) with bias 0; an unknown type
    raises NotImplementedError.  The parameters are views of the library's flat buffers: the values land there."""
    kind, gain = net.cfg["weight_init_type"], net.cfg["init_gain"]
    for m in net.modules():
        name = type(m).__name__
        if "BatchNorm" in name:
            torch.nn.init.normal_(m.weight.data, 1.0, gain)
            torch.nn.init.constant_(m.bias.data, 0.0)
        elif getattr(m, "weight", None) is not None and ("Linear" in name or "Conv" in name):
            if kind not in _WEIGHT_INIT:
                raise NotImplementedError('initialization method [%s] is not implemented' % kind)
            _WEIGHT_INIT[kind](m.weight.data, gain)


class DexGanGrasp(object):
    """Wrapper which houses the network blocks, the losses and training logic (dexgangrasp.py:65-823)."""

    def __init__(self, cfg):
        self.dtype = torch.float32
        torch.set_default_dtype(self.dtype)
        self.cfg = cfg
        self.is_train = cfg["is_train"]
        self.is_wgan = cfg["is_wgan"]
        world = parallel.env_rank_world()[1]
        if torch.cuda.is_available():
            if world > 1:
                # one process per GPU under torchrun: join the process group and make this rank's GPU the current
                # device before anything touches HIP (kernels launch on the current device)
                _, _, idx = parallel.init_from_env()
                self.device = torch.device('cuda:{}'.format(idx))
            else:
                self.device = torch.device('cuda:{}'.format(cfg["gpu_ids"][0]))
                torch.cuda.set_device(self.device)
        else:
            self.device = torch.device('cpu')
        if len(cfg["gpu_ids"]) > 1 and world == 1:
            raise NotImplementedError(
                "multi-GPU runs use one process per GPU: launch with torchrun / torch.distributed.run "
                "(gradients are all-reduced over RCCL); nn.DataParallel is not used")

        if cfg["model"] in ("ffhnet", "dexgangrasp"):
            self.DexGANGrasp, self.DexEvaluator = build_network(cfg, self.device, is_train=self.is_train)
        else:
            raise ValueError('Wrong configure model name of', cfg["model"])
        self._sync_replicas()   # every rank draws its own init (dexgangrasp.py:39-62): rank 0's wins

        if self.is_train:
            self.bce_weight = cfg["bce_weight"]
            self.kl_coef = cfg["kl_coef"]
            self.transl_coef = 100.
            self.rot_coef = 1.
            self.conf_coef = 10.
            self.train_dexgenerator = cfg["train_ffhgenerator"]
            self.train_dexevaluator = cfg["train_ffhevaluator"]
            adam = lambda params, lr: torch.optim.Adam(params, lr=lr, betas=(cfg["beta1"], 0.999),
                                                       weight_decay=cfg["weight_decay"])
            # The torch optimiser objects carry hyper-parameters, schedulers and the checkpoint format
            # (dexgangrasp.py:97-116); the update itself is the fused dgg_adam_l2 kernel.
            self.optim_dexgenerator = adam(self.DexGANGrasp.parameters(), cfg["lr"])
            self.optim_dexgangrasp_generator = adam(self.DexGANGrasp.generator.parameters(), cfg["lr_gen"])
            self.optim_dexgangrasp_discriminator = adam(self.DexGANGrasp.discriminator.parameters(), cfg["lr_dis"])
            self.optim_dexevaluator = adam(self.DexEvaluator.parameters(), cfg["lr"])
            self.scheduler_dexgenerator = networks.get_scheduler(self.optim_dexgenerator, cfg)
            self.scheduler_dexgangrasp_generator = networks.get_scheduler(self.optim_dexgangrasp_generator, cfg)
            self.scheduler_dexgangrasp_discriminator = networks.get_scheduler(self.optim_dexgangrasp_discriminator, cfg)
            self.scheduler_dexevaluator = networks.get_scheduler(self.optim_dexevaluator, cfg)
            self.estop_dexgenerator = EarlyStopping()
            self.estop_dexevaluator = EarlyStopping()

        self.kl_loss, self.rec_pose_loss = define_losses('transl_rot_6D_l2')
        self.L2_loss = torch.nn.MSELoss(reduction='mean')
        self.BCE_loss = torch.nn.BCELoss(reduction='mean')
        self.compute_eva_accuracy = losses.accuracy_evaluator

        dexgenerator_n_params = sum(p.numel() for p in self.DexGANGrasp.parameters() if p.requires_grad)
        print("The dexgenerator has {:2.2f} parms".format(dexgenerator_n_params))
        dexevaluator_n_params = sum(p.numel() for p in self.DexEvaluator.parameters() if p.requires_grad)
        print("The dexevaluator has {:2.2f} parms".format(dexevaluator_n_params))
        self.file_path = os.path.dirname(os.path.abspath(__file__))
        self.logit_thresh = 0.5
        self._engine = None
        self._z_override = None      # tests: inject the latent instead of torch.randn (:770)
        self._drop_keep = None       # tests: explicit evaluator dropout masks
        self._alpha_override = None  # tests: inject the WGAN-GP interpolation weights (B,3) instead of torch.randn (:151-153)

    # ------------------------------------------------------------------ engine plumbing
    def _sync_replicas(self, with_optim=False):
        """Data-parallel replicas must be identical: broadcast rank 0's parameters and BatchNorm
        statistics (after init and after every load_*; with_optim: Adam moments and step too)."""
        if not parallel.is_distributed() or not self.device.type == 'cuda':
            return
        for mod in (self.DexGANGrasp.generator, self.DexGANGrasp.discriminator, self.DexEvaluator):
            flat = mod._require_flat()
            bufs = [flat.params, flat.bn] + ([flat.m, flat.v] if with_optim and flat.m is not None else [])
            for t in bufs:
                parallel.broadcast_(t)
            if with_optim:
                st = torch.tensor([flat.step], dtype=torch.int64, device=self.device)
                parallel.broadcast_(st)
                flat.step = int(st.item())
            # (the counters are host tensors, networks._pin_bn_counters: one device vector carries them all)
            bns = [m for m in mod.modules() if isinstance(m, torch.nn.BatchNorm1d)]
            cnt = torch.tensor([int(m.num_batches_tracked) for m in bns], dtype=torch.int64, device=self.device)
            parallel.broadcast_(cnt)
            for m, v in zip(bns, cnt.cpu()):
                m.num_batches_tracked.fill_(int(v))

    def _eng(self):
        if self._engine is None:
            g = self.DexGANGrasp.generator._require_flat()
            d = self.DexGANGrasp.discriminator._require_flat()
            e = self.DexEvaluator._require_flat()
            self._engine = TrainEngine(self.cfg, g, d, e, self.device)
        else:  # modules may have been moved (cpu() for a checkpoint and back): re-resolve the buffers
            self._engine.G = self.DexGANGrasp.generator._require_flat()
            self._engine.D = self.DexGANGrasp.discriminator._require_flat()
            self._engine.E = self.DexEvaluator._require_flat()
        return self._engine

    def _flush(self):
        """Before anything reads the networks' state outside the train steps (evaluation forwards, inference,
        checkpoints): join the engine's side streams and drop a look-ahead pass that is still in flight
        (TrainEngine.flush) — the state is then the unpipelined order's."""
        if self._engine is not None:
            self._engine.flush()

    @staticmethod
    def _bump(mod, times):
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.num_batches_tracked += times

    def _export_adam(self, optim, module, flat):
        """Expose the fused Adam moments as a torch optimizer state (checkpoint format of
        dexgangrasp.py:662-712): state[p] = {step, exp_avg, exp_avg_sq}."""
        if flat is None or flat.step == 0:
            return
        mv, vv = flat.views("m"), flat.views("v")
        for name, p in module.named_parameters():
            optim.state[p] = {"step": torch.tensor(float(flat.step)), "exp_avg": mv[name], "exp_avg_sq": vv[name]}

    def _import_adam(self, optim, module, flat):
        """After optim.load_state_dict: copy loaded moments into the flat buffers the kernel uses."""
        mv, vv = flat.views("m"), flat.views("v")
        step = 0
        for name, p in module.named_parameters():
            st = optim.state.get(p)
            if st:
                mv[name].copy_(st["exp_avg"].to(mv[name].device))
                vv[name].copy_(st["exp_avg_sq"].to(vv[name].device))
                step = max(step, int(float(st["step"])))
        flat.step = step

    # ------------------------------------------------------------------ losses (torch-level API)
    def compute_loss_dexevaluator(self, pred_success_p):
        """dexgangrasp.py:142-147 (BCE in dgg_bce)."""
        loss, _ = ops.bce(pred_success_p.detach(), self.DexEvaluator.gt_label, weight=self.bce_weight, want_grad=False)
        return loss, {'total_loss_eva': loss, 'bce_loss': loss}

    def compute_loss_dexgangrasp_discriminator(self, real_score, fake_score, real_data=None, fake_data_disc=None):
        """dexgangrasp.py:213-232."""
        lr, _ = ops.bce(real_score.detach(), 1.0, weight=self.bce_weight, want_grad=False)
        lf, _ = ops.bce(fake_score.detach(), 0.0, weight=self.bce_weight, want_grad=False)
        tot = (lr + lf) / 2
        return tot, {'total_loss_disc': tot, 'bce_loss_real': lr, 'bce_loss_fake': lf}

    def compute_loss_dexgangrasp_generator(self, real_data, fake_data, fake_score):
        """dexgangrasp.py:272-307 (+ losses.py:61-76) evaluated by dgg_gen_loss (values only)."""
        dev = self.device
        nj = self.cfg["n_hand_joints"]
        B = fake_score.shape[0]
        heads = torch.cat([fake_data['joint_conf'].detach().float(), fake_data['rot_6D'].detach().float(),
                           fake_data['transl'].detach().float()], 1).contiguous()
        gt_rot = _lib.f32c(real_data['rot_matrix'], dev).reshape(B, 9)
        gt_tr = _lib.f32c(real_data['transl'], dev)
        gt_j = _lib.f32c(real_data['joint_conf'], dev)
        out = torch.zeros(16, dtype=torch.float32, device=dev)
        hp = ops.make_hparams(self.cfg)
        hp.bce_weight = float(self.bce_weight) if self.is_train else float(self.cfg.get("bce_weight", 1))
        ccfg = _lib.make_cfg(self.cfg)
        _lib.check(_lib.lib().dgg_gen_loss(_lib.C.byref(ccfg), _lib.C.byref(hp), B, _lib.ptr(heads), _lib.ptr(gt_rot),
                                           _lib.ptr(gt_tr), _lib.ptr(gt_j), _lib.ptr(_lib.f32c(fake_score.detach(), dev)),
                                           None, _lib.ptr(out), None, _lib.stream_ptr(dev)), "dgg_gen_loss")
        loss_dict = {'gen_loss_fake': out[_lib.L_GEN_FAKE], 'transl_loss': out[_lib.L_TRANSL],
                     'rot_loss': out[_lib.L_ROT], 'conf_loss': out[_lib.L_CONF],
                     'total_loss_gen': out[_lib.L_TOTAL_GEN]}
        self.last_loss_dict_gen = loss_dict
        return out[_lib.L_TOTAL_GEN], loss_dict

    # ------------------------------------------------------------------ eval helpers
    def eval_dexevaluator_accuracy(self, data):
        """dexgangrasp.py:309-322."""
        self._flush()
        with torch.no_grad():
            logits = self.DexEvaluator(data)
        pred_label = utils.class_labels_from_logits(logits.clone(), self.logit_thresh)
        pos_acc, neg_acc, acc = self.compute_eva_accuracy(pred_label, self.DexEvaluator.gt_label)
        return pos_acc, neg_acc, pred_label.detach().cpu().numpy(), self.DexEvaluator.gt_label.detach().cpu().numpy()

    def eval_dexevaluator_loss(self, data):
        """dexgangrasp.py:325-332."""
        self._flush()
        self.DexEvaluator.eval()
        with torch.no_grad():
            out = self.DexEvaluator(data)
            if not self.is_train:
                self.bce_weight = self.cfg["bce_weight"]
            _, loss_dict_dexevaluator = self.compute_loss_dexevaluator(out)
        return loss_dict_dexevaluator

    def eval_dexgangrasp_generator_loss(self, real_data):
        """dexgangrasp.py:334-365."""
        self._flush()
        self.DexGANGrasp.eval()
        with torch.no_grad():
            n_samples = real_data["bps_object"].shape[0]
            Zgen = self._draw_z(n_samples)
            y_fake = self.DexGANGrasp.generator(Zgen, real_data["bps_object"].to(self.DexGANGrasp.device))
            y_fake["rot_matrix"] = utils.rot_matrix_from_ortho6d(y_fake["rot_6D"])
            fake_data = {"bps_object": real_data["bps_object"], "rot_matrix": y_fake["rot_matrix"],
                         "rot_6D": y_fake["rot_6D"], "transl": y_fake["transl"], "joint_conf": y_fake["joint_conf"]}
            fake_score_gen = self.DexGANGrasp.discriminator(fake_data)
            real_data["rot_matrix"] = real_data["rot_matrix"].view(real_data["bps_object"].shape[0], -1)
            if not self.is_train:
                self.bce_weight = self.cfg["bce_weight"]
            _, loss_dict_dexgenerator = self.compute_loss_dexgangrasp_generator(real_data, fake_data, fake_score_gen)
        return loss_dict_dexgenerator

    # ------------------------------------------------------------------ inference
    def evaluate_grasps(self, bps, grasps, thresh=0.5, return_arr=True):
        """dexgangrasp.py:367-391."""
        self._flush()
        n_samples = grasps['rot_matrix'].shape[0]
        if len(bps.shape) > 1:
            bps = bps.squeeze()
        bps = np.tile(bps, (n_samples, 1))
        grasps['bps_object'] = bps
        grasps_t = utils.dict_to_tensor(grasps, device=self.device, dtype=self.dtype)
        with torch.no_grad():
            p_success = self.DexEvaluator(grasps_t).squeeze()
        if return_arr:
            p_success = p_success.cpu().detach().numpy()
        return p_success

    def _filter(self, scorer, bps, grasps, thresh, return_arr):
        n_samples = grasps['rot_matrix'].shape[0]
        if len(bps.shape) > 1:
            bps = bps.squeeze()
        bps = np.tile(bps, (n_samples, 1))
        grasps['bps_object'] = bps
        grasps_t = utils.dict_to_tensor(grasps, device=self.device, dtype=self.dtype)
        with torch.no_grad():
            p_success = scorer(grasps_t).squeeze()
        sorted_score, indices = p_success.sort(descending=True)
        if sorted_score[0] < thresh:
            raise ValueError("In total predicted ", n_samples, " grasps, but best score ", sorted_score[0],
                             " is still lower than thresh", thresh)
        indices = indices[sorted_score > thresh]
        filt_grasps = {k: torch.index_select(v, 0, indices) for k, v in grasps_t.items()}
        if return_arr:
            filt_grasps = {k: v.cpu().detach().numpy() for k, v in filt_grasps.items()}
        return filt_grasps

    def filter_grasps_multi(self, bps, grasps, thresholds=(0.5, 0.75, 0.90), return_arr=True, scorer=None):
        """The three-stage filter of dexgangrasp_offline.py:27-92 / eval.py:139-204 — filter_grasps called once
        per threshold on the SAME grasps — with ONE evaluator pass: the reference re-tiles the BPS row and
        re-runs the evaluator for every threshold (dexgangrasp.py:406-411), the scores are identical each time.
        Returns a list with, per threshold, the filtered dict (sorted by descending score, entries as
        filter_grasps returns them) or None where filter_grasps would raise ValueError (best score < thresh)."""
        self._flush()
        scorer = scorer or self.DexEvaluator
        n_samples = grasps['rot_matrix'].shape[0]
        if len(bps.shape) > 1:
            bps = bps.squeeze()
        grasps = dict(grasps)
        grasps['bps_object'] = np.tile(bps, (n_samples, 1))
        grasps_t = utils.dict_to_tensor(grasps, device=self.device, dtype=self.dtype)
        with torch.no_grad():
            p_success = scorer(grasps_t).squeeze(-1)
        sorted_score, indices = p_success.sort(descending=True)
        out = []
        for thresh in thresholds:
            if sorted_score[0] < thresh:
                out.append(None)
                continue
            idx = indices[sorted_score > thresh]
            f = {k: torch.index_select(v, 0, idx) for k, v in grasps_t.items()}
            if return_arr:
                f = {k: v.cpu().detach().numpy() for k, v in f.items()}
            out.append(f)
        return out

    def sweep_objects_sharded(self, bps_objects, n_samples, thresholds=(0.5, 0.75, 0.90), z_offset=0.025, z=None,
                              chunk_objects=256):
        """Config 5 across ranks ("replicas only, no collective; gather of filtered grasp counts at the
        end"; the per-object loop of dexgangrasp_offline.py:147-162 split by object).  Every rank calls this with the
        SAME list of objects; rank r sweeps the contiguous shard [r*ceil(O/W), ...) with its replica, then the
        per-threshold counts are all-gathered so every rank returns `count` (T, O) for ALL objects.  The grasps
        themselves stay on the rank that generated them: returns (local_result, (lo, hi), count_all)."""
        bps_objects = torch.as_tensor(bps_objects, dtype=self.dtype)
        n_obj = bps_objects.shape[0]
        w, r = parallel.world_size(), parallel.rank()
        per = (n_obj + w - 1) // w
        lo, hi = min(n_obj, r * per), min(n_obj, (r + 1) * per)
        zc = None if z is None else z[lo * n_samples:hi * n_samples]
        if hi > lo:
            local = self.sweep_objects(bps_objects[lo:hi], n_samples, thresholds, z_offset, zc, chunk_objects)
            cnt = local['count']
        else:
            local = None
            cnt = torch.zeros((len(thresholds), 0), dtype=torch.int64, device=self.device)
        gathered = parallel.gather_counts(cnt.t().contiguous())
        return local, (lo, hi), gathered.t().contiguous()

    def sweep_objects(self, bps_objects, n_samples, thresholds=(0.5, 0.75, 0.90), z_offset=0.025, z=None,
                      chunk_objects=256, dense=False):
        """Config 5 (BASELINE.json; dexgangrasp_offline.py:27-92 run over many objects) as ONE generator pass and ONE
        evaluator pass for all objects: the reference loops per object (generate_grasps -> filter_grasps x3), i.e.
        2 + 3 network passes of 200 rows each per object — launch-bound on any GPU.  Both networks are in eval mode
        here (running BatchNorm statistics), so rows of different objects do not interact and stacking them changes
        nothing.  The passes are FACTORISED per object (dgg_gen_forward_objects / dgg_eva_forward_objects): the BPS
        row is identical for an object's n samples, so its share of every first-layer product is computed once per
        object and the (objects x n) x 4096 tiled input is never built (dense=True keeps the tiled form, for tests).
        bps_objects (O, 4096); z: optional latents (O * n_samples, latentD) instead of torch.randn.
        Returns device tensors: 'transl' (O,n,3), 'rot_matrix' (O,n,3,3), 'joint_conf' (O,n,nj), 'z' (O,n,latentD),
        'score' (O,n), 'order' (O,n: indices by descending score) and 'count' (T,O): grasps with score > threshold —
        object o's filter_grasps(thresholds[t]) result is rows order[o, :count[t, o]].  Objects are processed
        chunk_objects at a time (activations of chunk_objects * n_samples rows live in the workspaces)."""
        self._flush()
        bps_objects = torch.as_tensor(bps_objects, dtype=self.dtype, device=self.device)
        if bps_objects.shape[0] > chunk_objects:
            parts = []
            for lo in range(0, bps_objects.shape[0], chunk_objects):
                hi = min(lo + chunk_objects, bps_objects.shape[0])
                zc = None if z is None else z[lo * n_samples:hi * n_samples]
                parts.append(self.sweep_objects(bps_objects[lo:hi], n_samples, thresholds, z_offset, zc, chunk_objects,
                                                dense))
            return {k: torch.cat([p[k] for p in parts], dim=1 if k == 'count' else 0) for k in parts[0]}
        gen, eva = self.DexGANGrasp, self.DexEvaluator
        gen.eval()
        eva.eval()
        n_obj = bps_objects.shape[0]
        nj = self.cfg["n_hand_joints"]
        with torch.no_grad():
            if z is None:
                z = torch.randn((n_obj * n_samples, gen.latentD), dtype=self.dtype, device=self.device)
            z = z.to(self.device, self.dtype).contiguous()
            if dense:
                bps_rows = bps_objects.repeat_interleave(n_samples, dim=0)
                res = gen.generator(z, bps_rows)
            else:
                heads = ops.gen_forward_objects(gen.generator._require_flat(), z, bps_objects, n_samples)
                res = {"rot_6D": heads[:, nj:nj + 6], "transl": heads[:, nj + 6:nj + 9], "joint_conf": heads[:, :nj], "z": z}
            rot = utils.rot_matrix_from_ortho6d(res.pop('rot_6D'))
            res['rot_matrix'] = rot
            res = utils.translate_along_axis(res, 0, z_offset)
            if dense:
                score = eva({'bps_object': bps_rows, 'rot_matrix': res['rot_matrix'], 'transl': res['transl']})
            else:
                score = ops.eva_forward_objects(eva._require_flat(), bps_objects, res['rot_matrix'].reshape(-1, 9),
                                                res['transl'], n_samples)
            score = score.view(n_obj, n_samples)
            sorted_score, order = score.sort(dim=1, descending=True)
            count = torch.stack([(sorted_score > t).sum(dim=1) for t in thresholds])
        out = {k: v.reshape(n_obj, n_samples, *v.shape[1:]) for k, v in res.items()}
        out.update(score=score, order=order, count=count)
        return out

    def filter_grasps(self, bps, grasps, thresh=0.5, return_arr=True):
        """dexgangrasp.py:393-442."""
        return self._filter(self.DexEvaluator, bps, grasps, thresh, return_arr)

    def filter_grasps_discriminator(self, bps, grasps, thresh=0.5, return_arr=True):
        """dexgangrasp.py:444-493."""
        return self._filter(self.DexGANGrasp.discriminator, bps, grasps, thresh, return_arr)

    def generate_grasps(self, bps, n_samples, return_arr=True, z_offset=0.025):
        """dexgangrasp.py:495-514."""
        self._flush()
        if len(bps.shape) > 1:
            bps = bps.squeeze()
        bps = np.tile(bps, (n_samples, 1))
        bps_tensor = torch.tensor(bps, dtype=self.dtype, device=self.device)
        with torch.no_grad():
            return self.DexGANGrasp.generate_poses(bps_tensor, return_arr=return_arr, z_offset=z_offset)

    # ------------------------------------------------------------------ N4: gradient-based refinement
    def improve_grasps_gradient_based(self, data, last_success):
        """dexgangrasp.py:516-541: one ascent step on the evaluator's success probability, each sample's step scaled
        so that the palm translation moves at most 1 cm.  The input gradients come from `dgg_eva_backward`
        (DGG_NEED_INPUT_GRADS) through the module's autograd bridge and — as in the reference, which never zeroes
        them — accumulate in `.grad` over the steps of one `refine_grasps` call.
        The reference also steps `joint_conf`, whose `.grad` is None because DexEvaluator does not read it
        (networks.py:307-338), i.e. its line 539 raises TypeError; here joint_conf is left unchanged."""
        p_success = self.DexEvaluator(data)
        p_success.squeeze(-1).backward(torch.ones(p_success.shape[0], device=p_success.device))
        norm_transl = torch.norm(data["transl"].grad, p=2, dim=-1)
        alpha = torch.min(0.01 / norm_transl, torch.tensor(1.0, device=norm_transl.device))
        data["transl"].data += data["transl"].grad * alpha[:, None]
        data["rot_matrix"].data += data["rot_matrix"].grad * alpha.view(-1, *([1] * (data["rot_matrix"].dim() - 1)))
        if data["joint_conf"].grad is not None:
            data["joint_conf"].data += data["joint_conf"].grad * alpha[:, None]
        return p_success.squeeze(), None

    def refine_grasps(self, data, refine_method, num_refine_steps=10, dtype=torch.float32):
        """dexgangrasp.py:619-660.  `refine_method == "gradient"` only: the sampling-based variant of the reference
        calls helpers that do not exist in its tree (`utils.control_points_from_rot_and_trans`, `self.grasp_evaluator`)."""
        self._flush()
        data = utils.data_dict_to_dtype(data, dtype)
        if refine_method != "gradient":
            raise NotImplementedError("refine method [%s] is not implemented" % refine_method)
        for k in ("rot_matrix", "transl", "joint_conf"):
            data[k] = data[k].to(self.device).detach().clone().requires_grad_(True)
        refined_success, refined_data = [], [utils.grasp_numpy_from_data_dict(data)]
        last_success = None
        for _ in range(num_refine_steps):
            p_success, last_success = self.improve_grasps_gradient_based(data, last_success)
            refined_success.append(p_success.cpu().data.numpy())
            refined_data.append(utils.grasp_numpy_from_data_dict(data))
        with torch.no_grad():
            refined_success.append(self.DexEvaluator(data).squeeze().cpu().data.numpy())
        return refined_data, refined_success

    # ------------------------------------------------------------------ checkpoints
    def load_dexevaluator(self, epoch, load_path=None):
        """dexgangrasp.py:571-592."""
        if load_path is None:
            load_path = self.cfg["load_path"]
        load_path = os.path.join(load_path, str(epoch) + '_eva_net.pt')
        ckpt = torch.load(load_path, map_location=self.device, weights_only=False)
        self.DexEvaluator.load_state_dict(ckpt['ffhevaluator_state_dict'])
        if self.cfg["is_train"]:
            self.optim_dexevaluator.load_state_dict(ckpt['optim_ffhevaluator_state_dict'])
            self.scheduler_dexevaluator.load_state_dict(ckpt['scheduler_ffhevaluator_state_dict'])
            self._import_adam(self.optim_dexevaluator, self.DexEvaluator, self.DexEvaluator._require_flat())
            self.cfg["load_epoch"] = ckpt['epoch']
            self.DexEvaluator.train()
        else:
            self.DexEvaluator.eval()
        self._sync_replicas(with_optim=self.cfg["is_train"])

    def load_dexgenerator(self, epoch, load_path=None):
        """dexgangrasp.py:594-617 (+ the GAN optimisers, which the reference saves but leaves as a TODO
        to reload at :611)."""
        if load_path is None:
            load_path = self.cfg["load_path"]
        load_path = os.path.join(load_path, str(epoch) + '_gen_net.pt')
        ckpt = torch.load(load_path, map_location=self.device, weights_only=False)
        self.DexGANGrasp.load_state_dict(ckpt['ffhgenerator_state_dict'])
        if self.cfg["is_train"]:
            print("Load TRAIN mode")
            self.optim_dexgenerator.load_state_dict(ckpt['optim_ffhgenerator_state_dict'])
            self.scheduler_dexgenerator.load_state_dict(ckpt['scheduler_ffhgenerator_state_dict'])
            for key, opt, mod in (("optim_ffhgan_generator_state_dict", self.optim_dexgangrasp_generator,
                                   self.DexGANGrasp.generator),
                                  ("optim_ffhgan_discriminator_state_dict", self.optim_dexgangrasp_discriminator,
                                   self.DexGANGrasp.discriminator)):
                if key in ckpt and ckpt[key].get("state"):
                    opt.load_state_dict(ckpt[key])
                    self._import_adam(opt, mod, mod._require_flat())
            self.DexGANGrasp.train()
        else:
            print("Network in EVAL mode")
            self.DexGANGrasp.eval()
        self._sync_replicas(with_optim=self.cfg["is_train"])

    def save_dexevaluator(self, net_name, epoch):
        """dexgangrasp.py:662-683: same file name and dict keys."""
        save_path = os.path.join(self.cfg["save_dir"], net_name + '_eva_net.pt')
        if self._engine is not None:
            self._engine.flush()
        self._export_adam(self.optim_dexevaluator, self.DexEvaluator, self.DexEvaluator._require_flat())
        sd = {k: v.detach().cpu().clone() for k, v in self.DexEvaluator.state_dict().items()}
        torch.save({'epoch': epoch, 'ffhevaluator_state_dict': sd,
                    'optim_ffhevaluator_state_dict': self.optim_dexevaluator.state_dict(),
                    'scheduler_ffhevaluator_state_dict': self.scheduler_dexevaluator.state_dict()}, save_path)

    def save_dexgenerator(self, net_name, epoch):
        """dexgangrasp.py:685-712: same file name and dict keys."""
        save_path = os.path.join(self.cfg["save_dir"], net_name + '_gen_net.pt')
        if self._engine is not None:
            self._engine.flush()
        self._export_adam(self.optim_dexgangrasp_generator, self.DexGANGrasp.generator,
                          self.DexGANGrasp.generator._require_flat())
        self._export_adam(self.optim_dexgangrasp_discriminator, self.DexGANGrasp.discriminator,
                          self.DexGANGrasp.discriminator._require_flat())
        sd = {k: v.detach().cpu().clone() for k, v in self.DexGANGrasp.state_dict().items()}
        torch.save({'epoch': epoch, 'ffhgenerator_state_dict': sd,
                    'optim_ffhgenerator_state_dict': self.optim_dexgenerator.state_dict(),
                    'scheduler_ffhgenerator_state_dict': self.scheduler_dexgenerator.state_dict(),
                    'optim_ffhgan_generator_state_dict': self.optim_dexgangrasp_generator.state_dict(),
                    'scheduler_ffhgan_generator_state_dict': self.scheduler_dexgangrasp_generator.state_dict(),
                    'optim_ffhgan_discriminator_state_dict': self.optim_dexgangrasp_discriminator.state_dict(),
                    'scheduler_ffhgan_discriminator_state_dict': self.scheduler_dexgangrasp_discriminator.state_dict(),
                    }, save_path)

    # ------------------------------------------------------------------ schedule hooks
    def update_estop(self, eval_loss_dict):
        """dexgangrasp.py:714-725."""
        if self.train_dexevaluator:
            if self.estop_dexevaluator(eval_loss_dict['total_loss_eva']):
                self.train_dexevaluator = False
        if self.train_dexgenerator:
            if self.estop_dexgenerator(eval_loss_dict['total_loss_gen']):
                self.train_dexgenerator = False

    def update_learning_rate(self, eval_loss_dict):
        """dexgangrasp.py:727-737."""
        if self.train_dexevaluator:
            self.scheduler_dexevaluator.step(eval_loss_dict['total_loss_eva'])
            print('learning rate evaluator = %.7f' % self.optim_dexevaluator.param_groups[0]['lr'])
        if self.train_dexgenerator:
            self.scheduler_dexgenerator.step(eval_loss_dict['total_loss_gen'])
            print('learning rate generator = %.7f' % self.optim_dexgenerator.param_groups[0]['lr'])

    # ------------------------------------------------------------------ training steps
    _Z_BLOCK = 32   # steps whose latents one torch.randn launch draws (same generator, same distribution: :770 per step)

    def _draw_z(self, n):
        """The latent of one update (dexgangrasp.py:770 draws torch.randn per step).  Drawn from the same device generator
        in blocks of _Z_BLOCK steps: one launch per block instead of one per step on the step's stream (part of the wrapper's
        overhead over the engine calls; tests inject theirs through _z_override)."""
        if self._z_override is not None:
            return self._z_override.to(self.device, self.dtype)
        blk = getattr(self, "_z_block", None)
        if blk is None or blk[0].shape[1] != n or blk[1] >= self._Z_BLOCK:
            blk = [torch.randn((self._Z_BLOCK, n, self.DexGANGrasp.latentD), dtype=self.DexGANGrasp.dtype,
                               device=self.DexGANGrasp.device), 0]
            self._z_block = blk
        z = blk[0][blk[1]]
        blk[1] += 1
        return z

    def _gen_losses(self, out, is_train_gen):
        """last_loss_dict_gen (:299-306 caches the generator's loss terms; :817-822 re-reports them while is_train_gen is
        False).  The entries are views of the step's loss vector, which lives in a ring (engine.TrainEngine.losses): the
        first call that RE-reports them takes a copy, so that they outlive the ring however long the generator pauses."""
        if is_train_gen:
            self.last_loss_dict_gen = {'gen_loss_fake': out[_lib.L_GEN_FAKE], 'transl_loss': out[_lib.L_TRANSL],
                                       'rot_loss': out[_lib.L_ROT], 'conf_loss': out[_lib.L_CONF],
                                       'total_loss_gen': out[_lib.L_TOTAL_GEN]}
            self._gen_losses_are_views = True
        elif getattr(self, "_gen_losses_are_views", False):
            self.last_loss_dict_gen = {k: v.clone() for k, v in self.last_loss_dict_gen.items()}
            self._gen_losses_are_views = False
        return self.last_loss_dict_gen

    def update_dexevaluator(self, data):
        """dexgangrasp.py:739-755 as one fused phase (dgg_eva_phase) + dgg_adam_l2."""
        self.DexEvaluator.train()
        eva = self.DexEvaluator
        eva.set_input(data)  # same casts / side-effect attributes as the reference's forward (:307-313)
        seed = 0
        if self._drop_keep is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        lr = self.optim_dexevaluator.param_groups[0]['lr']
        out = self._eng().eva_step(eva.bps_object, eva.rot_matrix, eva.transl, eva.gt_label, self._drop_keep, seed, lr,
                                   two_streams=_API_STREAMS)
        self._bump(eva, 1)
        return {'total_loss_eva': out[_lib.L_TOTAL_EVA], 'bce_loss': out[_lib.L_BCE_EVA]}

    def update_joint(self, real_data, eva_data, is_train_gen=True, next_real_data=None):
        """update_dexgangrasp(real_data, is_train_gen) and update_dexevaluator(eva_data) as ONE interleaved step
        (engine.joint_step): the two updates touch disjoint networks, batches and optimisers, so the evaluator's runs
        beside the cGAN's on a second stream — same parameters afterwards, bit for bit, as the two calls one after the
        other (train.py:249-276 runs them in two epoch loops).  next_real_data: the NEXT step's generator batch when it
        is already resident on the device as contiguous float32 tensors (a look-ahead loader): its discriminator
        real-batch pass is then enqueued one step ahead.  Returns (cGAN loss dict, evaluator loss dict).  The latent and
        the dropout seed are drawn exactly as the two separate calls draw them (Z first)."""
        if self.is_wgan:
            return self.update_dexgangrasp(real_data, is_train_gen), self.update_dexevaluator(eva_data)
        gan, eva = self.DexGANGrasp, self.DexEvaluator
        gan.train()
        eva.train()
        dev = gan.device
        n = real_data["bps_object"].shape[0]
        Zgen = self._draw_z(n)
        g_in = (_lib.f32c(real_data["bps_object"], dev), _lib.f32c(real_data["rot_matrix"], dev).reshape(n, 9),
                _lib.f32c(real_data["transl"], dev), _lib.f32c(real_data["joint_conf"], dev))
        eva.set_input(eva_data)
        seed = 0
        if self._drop_keep is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        nxt = None
        if next_real_data is not None and is_train_gen and next_real_data["bps_object"].shape[0] == n:
            cand = (next_real_data["bps_object"], next_real_data["rot_matrix"].reshape(n, 9), next_real_data["transl"],
                    next_real_data["joint_conf"])
            if all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in cand):
                nxt = cand
        out = self._eng().joint_step(g_in, Zgen, (eva.bps_object, eva.rot_matrix, eva.transl, eva.gt_label), is_train_gen,
                                     self._drop_keep, seed, self.optim_dexgangrasp_generator.param_groups[0]['lr'],
                                     self.optim_dexgangrasp_discriminator.param_groups[0]['lr'],
                                     self.optim_dexevaluator.param_groups[0]['lr'], next_gan_inputs=nxt)
        self._bump(gan.generator, 1)
        self._bump(gan.discriminator, 2 + (1 if is_train_gen else 0))
        self._bump(eva, 1)
        loss_dict = {'total_loss_disc': out[_lib.L_TOTAL_DISC], 'bce_loss_real': out[_lib.L_BCE_REAL],
                     'bce_loss_fake': out[_lib.L_BCE_FAKE]}
        if is_train_gen:
            real_data["rot_matrix"] = real_data["rot_matrix"].view(n, -1)  # :810 mutates the caller's dict
        loss_dict.update(self._gen_losses(out, is_train_gen))
        return loss_dict, {'total_loss_eva': out[_lib.L_TOTAL_EVA], 'bce_loss': out[_lib.L_BCE_EVA]}

    def update_dexgangrasp(self, real_data, is_train_gen=True):
        """dexgangrasp.py:757-823: discriminator phase, Adam(D), generator phase, Adam(G) — the 13
        steps of SURVEY.md 3.1 — as two fused phases of libdgg.so."""
        gan = self.DexGANGrasp
        gan.train()
        dev = gan.device
        n_samples = real_data["bps_object"].shape[0]
        Zgen = self._draw_z(n_samples)
        bps = _lib.f32c(real_data["bps_object"], dev)
        rot9 = _lib.f32c(real_data["rot_matrix"], dev).reshape(n_samples, 9)
        transl = _lib.f32c(real_data["transl"], dev)
        joint = _lib.f32c(real_data["joint_conf"], dev)
        lr_g = self.optim_dexgangrasp_generator.param_groups[0]['lr']
        lr_d = self.optim_dexgangrasp_discriminator.param_groups[0]['lr']
        if self.is_wgan:
            # :149-158 calculate_interp draws three torch.randn weights per sample: rot (B,1,1), transl (B,1), joint (B,1)
            if self._alpha_override is not None:
                alphas = self._alpha_override.to(dev, self.dtype).reshape(n_samples, 3)
            else:
                alphas = torch.cat([torch.randn((n_samples, 1, 1), device=dev).view(n_samples, 1),
                                    torch.randn((n_samples, 1), device=dev), torch.randn((n_samples, 1), device=dev)], 1)
            out = self._eng().wgan_step(bps, rot9, transl, joint, Zgen, alphas, is_train_gen, lr_g, lr_d)
        else:
            out = self._eng().cgan_step(bps, rot9, transl, joint, Zgen, is_train_gen, lr_g, lr_d, two_streams=_API_STREAMS)
        self._bump(gan.generator, 1)
        # D forwards per step: real, fake [, interpolates (WGAN-GP)] [, generator phase]
        self._bump(gan.discriminator, 2 + (1 if self.is_wgan else 0) + (1 if is_train_gen else 0))
        loss_dict = {'total_loss_disc': out[_lib.L_TOTAL_DISC], 'bce_loss_real': out[_lib.L_BCE_REAL],
                     'bce_loss_fake': out[_lib.L_BCE_FAKE]}
        if is_train_gen:
            real_data["rot_matrix"] = real_data["rot_matrix"].view(n_samples, -1)  # :810 mutates the caller's dict
        loss_dict.update(self._gen_losses(out, is_train_gen))
        return loss_dict
