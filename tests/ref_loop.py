"""TEST INFRASTRUCTURE - the reference's training loop body, restated (reference engine/engine.py:37-73, utils/misc.py:114-129).

The reference drives its model through `engine.train(train_loader, model, optimizer, scheduler, scaler, epoch, args)`.  That
function cannot travel to the GPU box (the reference does not exist there) and needs wandb / loguru / a DataLoader, so the
GPU tests use this restatement of its per-batch body; tests/test_ref_loop_cpu.py runs BOTH - the reference's own function,
imported from /root/reference, and this one - on the same small CPU model and requires bit-identical parameters afterwards,
which pins the restatement to the reference.  What is left out is logging only (AverageMeter / ProgressMeter / wandb)."""
import torch
import torch.distributed as dist


def train_metric_gpu(output, target, threshold=0.35, pr_iou=0.5):
    """utils/misc.py:114-129"""
    assert output.dim() in [2, 3, 4]
    assert output.shape == target.shape
    output = output.flatten(1)
    target = target.flatten(1)
    output = torch.sigmoid(output)
    output[output < threshold] = 0.
    output[output >= threshold] = 1.
    inter = (output.bool() & target.bool()).sum(dim=1)
    union = (output.bool() | target.bool()).sum(dim=1)
    ious = inter / (union + 1e-6)
    iou = ious.mean()
    prec = (ious > pr_iou).float().mean()
    return 100. * iou, 100. * prec


def train_steps(batches, model, optimizer, scaler, max_norm=0.0, device_type="cuda"):
    """engine/engine.py:29-73 without the meters: model.train(); per batch - move to the device (mask gets its channel
    dimension, :42), forward under ambient autocast (:48-49), zero_grad, scaled backward, optional clipping, scaler.step,
    scaler.update (:52-57), trainMetricGPU and the three scalar all-reduces averaged over the world (:60-66).
    Returns [(loss, iou, pr5)] as floats (what the reference feeds its meters)."""
    model.train()
    out = []
    for image, text, target in batches:
        if device_type == "cuda":
            image = image.cuda(non_blocking=True)
            text = text.cuda(non_blocking=True)
            target = target.cuda(non_blocking=True).unsqueeze(1)
        else:
            target = target.unsqueeze(1)
        with torch.autocast(device_type, enabled=device_type == "cuda"):
            pred, target, loss = model(image, text, target)
        optimizer.zero_grad()
        scaler.scale(loss).backward()
        if max_norm:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        scaler.step(optimizer)
        scaler.update()
        iou, pr5 = train_metric_gpu(pred, target, 0.35, 0.5)
        dist.all_reduce(loss.detach())
        dist.all_reduce(iou)
        dist.all_reduce(pr5)
        world = dist.get_world_size()
        out.append((float(loss) / world, float(iou) / world, float(pr5) / world))
    return out
