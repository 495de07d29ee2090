"""Checkpoint save / resume for the training loop, with the sparse-training state the reference leaves out.

Mirrors ``utils.save_model`` (utils.py:447-469) and ``utils.auto_load_model1`` (utils.py:470-512): same file name pattern
(``checkpoint-<epoch>.pth``), same keys (``model``, ``optimizer``, ``epoch``, ``scaler``, ``args``, ``model_ema``), same pruning of
old checkpoints -- plus ``mask``: ``Masking.state_dict()`` (bit-packed masks, step counter, prune-rate schedule position), restored
by ``auto_load_model`` so that a resumed run continues with the SAME masks instead of re-deriving them as ``weight != 0``
(``--sparse_init resume``, sparse_core.py:158-172).
"""
import glob
import os
from pathlib import Path

import torch


def _is_main_process():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, model_ema=None, mask=None):
    output_dir = Path(args.output_dir)
    path = output_dir / ('checkpoint-%s.pth' % str(epoch))
    to_save = {'model': model_without_ddp.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch,
               'scaler': loss_scaler.state_dict() if loss_scaler is not None else None, 'args': args}
    if model_ema is not None:
        to_save['model_ema'] = model_ema.ema.state_dict()
    if mask is not None:
        to_save['mask'] = mask.state_dict()
    if _is_main_process():
        torch.save(to_save, path)
        if isinstance(epoch, int):
            old = output_dir / ('checkpoint-%s.pth' % (epoch - args.save_ckpt_num * args.save_ckpt_freq))
            if os.path.exists(old):
                os.remove(old)


def auto_load_model(args, model, model_without_ddp, optimizer, loss_scaler, model_ema=None, mask=None):
    """Resume from ``args.resume`` or, with ``args.auto_resume``, from the newest ``checkpoint-<int>.pth`` in ``args.output_dir``.
    Unlike the reference (whose optimizer/epoch restore is commented out: utils.py:494-499) the optimizer, epoch, scaler, EMA and
    mask state are all restored when present."""
    output_dir = Path(args.output_dir)
    if getattr(args, 'auto_resume', False) and len(getattr(args, 'resume', '') or '') == 0:
        latest = -1
        for ckpt in glob.glob(os.path.join(output_dir, 'checkpoint-*.pth')):
            t = ckpt.split('-')[-1].split('.')[0]
            if t.isdigit():
                latest = max(int(t), latest)
        if latest >= 0:
            args.resume = os.path.join(output_dir, 'checkpoint-%d.pth' % latest)
        print("Auto resume checkpoint: %s" % args.resume)
    if not getattr(args, 'resume', ''):
        return False
    checkpoint = torch.load(args.resume, map_location='cpu', weights_only=False)
    model_without_ddp.load_state_dict(checkpoint['model'])
    print("Resume checkpoint %s" % args.resume)
    if optimizer is not None and 'optimizer' in checkpoint:
        optimizer.load_state_dict(checkpoint['optimizer'])
    if 'epoch' in checkpoint and not isinstance(checkpoint['epoch'], str):
        args.start_epoch = checkpoint['epoch'] + 1
    if loss_scaler is not None and checkpoint.get('scaler') is not None:
        loss_scaler.load_state_dict(checkpoint['scaler'])
    if model_ema is not None and 'model_ema' in checkpoint:
        model_ema.ema.load_state_dict(checkpoint['model_ema'])
    if mask is not None and 'mask' in checkpoint:
        mask.load_state_dict(checkpoint['mask'])
    return True
