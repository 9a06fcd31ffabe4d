"""Mirror of the live part of DexGanGrasp/models/losses.py: accuracy_evaluator (:5-16) and
transl_rot_6D_l2_loss (:61-76, the reconstruction loss selected at dexgangrasp.py:120).  The
fused training step computes these inside dgg_gen_loss; the functions below serve callers that
use them directly (they operate on device tensors; ortho6d runs in libdgg.so)."""
import torch

from ..utils import utils


def accuracy_evaluator(pred_label, gt_label):
    correct = torch.eq(pred_label, gt_label)
    true_positive = torch.sum(correct * gt_label) / torch.sum(gt_label)
    true_negative = torch.sum(correct * (1. - gt_label)) / torch.sum(1. - gt_label)
    acc = correct.sum() / float(len(gt_label))
    return true_positive, true_negative, acc


def transl_rot_6D_l2_loss(pred_transl_rot_6D, gt_transl_rot_matrix, torch_l2_loss_fn, device='cpu'):
    pred_rot_matrix = utils.rot_matrix_from_ortho6d(pred_transl_rot_6D['rot_6D'])
    pred_rot_matrix = pred_rot_matrix.reshape(pred_rot_matrix.shape[0], -1)
    gt_rot_matrix = gt_transl_rot_matrix['rot_matrix']
    l2_loss_rot = torch_l2_loss_fn(pred_rot_matrix, gt_rot_matrix)
    l2_loss_transl = torch_l2_loss_fn(pred_transl_rot_6D['transl'], gt_transl_rot_matrix['transl'])
    return l2_loss_transl, l2_loss_rot
