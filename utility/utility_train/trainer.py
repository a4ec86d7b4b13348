"""Epoch loop shared by every model plugin — `universal_trainer`, same signature, same
console / log lines and same stopping rule as the reference's
utility/utility_train/trainer.py:8-74.

Per epoch: draw one negative per train edge (native sampler on NumPy's global stream), move
the triples to the device, shuffle them with the same stream, walk the mini-batches.  Two
ways to run a step:
  * generic: `model(batch) -> [losses]`, sum, autograd backward, optimizer step — works for
    any nn.Module built from the differentiable operators in idgrec_amd.ops;
  * fused (models that set `supports_fused_step`): `model.fused_train_step(batch, optimizer)` runs
    forward, backward and the Adam update as one fixed chain of HIP kernels, no autograd graph
    (`model.fused_loss_and_grad(batch)` + `optimizer.step()` when the optimizer is not ours).
Either way the per-step `loss.item()` host round trip of the reference (trainer.py:52) is
replaced by one device->host copy per epoch; the logged numbers are formed from the same
fp32 per-step losses, accumulated in float64 in step order, as the reference's Python floats.
"""
from time import time

import torch
from tqdm import tqdm

import utility.utility_function.tools as tools
import utility.utility_train.batch_test as batch_test
from idgrec_amd import ops


def _make_optimizer(model, lr, device):
    params = list(model.parameters())
    if all(p.is_cuda and p.dtype == torch.float32 for p in params):
        return ops.Adam(params, lr=lr)  # torch.optim.Adam's default algorithm as one HIP kernel per tensor
    return torch.optim.Adam(params, lr=lr)


def universal_trainer(model, args, config, dataset, device, logger):
    model.to(device)
    batch_size = int(config['batch_size'])
    top_k = eval(config['top_K'])
    Optim = _make_optimizer(model, float(config['learn_rate']), device)
    # the fused chain exists for the tiled embedding widths only: models say whether it applies (embedding_size = 48
    # and the like fall back to forward() + autograd + the same Adam kernel)
    available = getattr(model, "fused_step_available", None)
    fused = torch.device(device).type == "cuda" and (available() if available is not None
                                                     else bool(getattr(model, "supports_fused_step", False)))

    best_results = {'count': 0, 'epoch': 0, 'recall': [0. for _ in top_k], 'ndcg': [0. for _ in top_k], 'stop': 0}

    def draw_epoch():
        """sample -> device -> shuffle, in the reference's order on NumPy's global stream (trainer.py:26-34)."""
        sample_data = dataset.sample_data_to_train_all()
        triples = torch.from_numpy(sample_data).to(device)  # int64 ids: no float32 round trip (trainer.py:27-29)
        u, p, n = tools.shuffle(triples[:, 0], triples[:, 1], triples[:, 2])
        return u.contiguous(), p.contiguous(), n.contiguous()

    n_epochs, interval = int(config['training_epochs']), int(config['interval'])
    on_gpu = torch.device(device).type == "cuda"
    lookahead = None  # the next epoch's triples, drawn while the device was still working through this one
    for epoch in range(n_epochs):
        print('-' * 100)
        start_time = time()
        model.train()

        users, pos_items, neg_items = lookahead if lookahead is not None else draw_epoch()
        lookahead = None
        num_batch = len(users) // batch_size + 1  # the reference's divisor, also when batch_size | E

        step_losses = None
        batches = list(tools.mini_batch(users, pos_items, neg_items, batch_size=batch_size))
        for batch_i, (b_users, b_pos, b_neg) in tqdm(enumerate(batches), desc='Training epoch ' + str(epoch + 1),
                                                       total=int(num_batch)):
            if fused:
                if step_losses is None:
                    step_losses = torch.zeros((num_batch, int(getattr(model, "n_fused_losses", 2))), dtype=torch.float32,
                                              device=device)
                if batch_i + 1 < len(batches):
                    model.prefetch_batch(*batches[batch_i + 1])  # index-only work of the next step, off the critical path
                # one chain of kernels for forward + backward + Adam when the optimizer is ours ...
                if not model.fused_train_step(b_users, b_pos, b_neg, step_losses[batch_i], Optim):
                    # ... otherwise gradients from the fused path, update by whatever optimizer this is
                    model.fused_loss_and_grad(b_users, b_pos, b_neg, loss_out=step_losses[batch_i])
                    Optim.step()
                continue
            loss_list = model(b_users, b_pos, b_neg)
            if step_losses is None:
                assert len(loss_list) >= 1
                step_losses = torch.zeros((num_batch, len(loss_list)), dtype=torch.float32, device=device)
            total_loss = 0.
            for i, loss in enumerate(loss_list):
                total_loss = total_loss + loss
                step_losses[batch_i, i] = loss.detach()
            Optim.zero_grad()
            total_loss.backward()
            Optim.step()

        # The steps above were only ISSUED: the device is still working through them.  Draw the next epoch's
        # negatives now (host work, ~35 ms at yelp2018 size) instead of after the copy below has waited for the
        # device.  Same draws in the same order — nothing else touches NumPy's stream in between; skipped when
        # this epoch's test could end the run (the reference would then never have drawn them).
        will_test = epoch % interval == 0
        may_stop = will_test and best_results['count'] + 1 >= int(config['early_stopping'])
        if on_gpu and epoch + 1 < n_epochs and not may_stop:
            lookahead = draw_epoch()

        # one host copy per epoch; float64 accumulation in step order == summing loss.item() per step
        per_step = step_losses.double().cpu().numpy() if step_losses is not None else []
        total_loss_list = [0.] * (per_step.shape[1] if len(per_step) else 0)
        for row in per_step:
            for i, v in enumerate(row):
                total_loss_list[i] += float(v)
        end_time = time()

        loss_strs = str(round(sum(total_loss_list) / num_batch, 6)) \
            + " = " + " + ".join([str(round(i / num_batch, 6)) for i in total_loss_list])
        print("Training time: %.3f | training loss: %s" % (end_time - start_time, loss_strs))
        logger.info("Epoch: %4d | Training time: %.3f | training loss: %s" % (epoch + 1, end_time - start_time, loss_strs))

        if will_test:
            result, best_results = batch_test.general_test(dataset, model, device, config, epoch, best_results)
            logger.info("Epoch: %4d | Test recall: %s | Test NDCG: %s" % (epoch + 1, result['recall'], result['ndcg']))
            if best_results['stop'] > 0:
                break

    print("Model training process completed.")
    logger.info('Model training process completed.')
    logger.info("Best epoch: %4d | Best recall: %s | Best NDCG: %s"
                % (best_results['epoch'], best_results['recall'], best_results['ndcg']))
