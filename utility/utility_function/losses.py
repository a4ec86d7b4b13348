"""Loss functions with the reference's names (utility/utility_function/losses.py).

`get_bpr_loss` / `get_reg_loss` operate on already-gathered [B, d] blocks exactly like the
reference and are plain torch expressions (they serve models that gather on their own, e.g.
NGCF-style encoders).  The LightGCN / MFBPR / SimGCL models do not come through here for
their main loss: they call `idgrec_amd.ops.bpr_loss`, which fuses gather + both losses +
gradients in one HIP kernel chain.
"""
import torch


def get_bpr_loss(user_embedding, positive_embedding, negative_embedding):
    x = (user_embedding * positive_embedding).sum(dim=1) - (user_embedding * negative_embedding).sum(dim=1)
    return torch.mean(-torch.log(torch.sigmoid(x) + 10e-8))  # epsilon is 1e-7, as in losses.py:11


def get_reg_loss(*embeddings):
    total = 0
    for block in embeddings:
        total = total + 1 / 2 * block.norm(2).pow(2) / float(block.shape[0])
    return total


def _cosine_logits(a, b, temperature):
    a = torch.nn.functional.normalize(a)
    b = torch.nn.functional.normalize(b)
    return a, b, torch.exp((a * b).sum(dim=-1) / temperature)


def get_InfoNCE_loss(embedding_1, embedding_2, temperature):
    """In-batch InfoNCE; note the 1e-5 guard (10e-6 in losses.py:34)."""
    a, b, pos = _cosine_logits(embedding_1, embedding_2, temperature)
    ttl = torch.exp(torch.matmul(a, b.transpose(0, 1)) / temperature).sum(dim=1)
    return torch.mean(-torch.log(pos / ttl + 10e-6))


def get_InfoNCE_loss_all(embedding_1, embedding_2, embedding_2_all, temperature):
    a, b, pos = _cosine_logits(embedding_1, embedding_2, temperature)
    every = torch.nn.functional.normalize(embedding_2_all)
    ttl = torch.exp(torch.matmul(a, every.transpose(0, 1)) / temperature).sum(dim=1)
    return torch.mean(-torch.log(pos / ttl + 10e-8))
