"""Episode tensor helpers (surface of test_phase/utils/few_shot.py:4-16)."""
import torch


def split_shot_query(data, way, shot, query, ep_per_batch=1):
    """data [E*way*(shot+query), ...] in class-major order -> x_shot [E,way,shot,...],
    x_query [E,way*query,...]."""
    img_shape = data.shape[1:]
    data = data.view(ep_per_batch, way, shot + query, *img_shape)
    x_shot, x_query = data.split([shot, query], dim=2)
    x_shot = x_shot.contiguous()
    x_query = x_query.contiguous().view(ep_per_batch, way * query, *img_shape)
    return x_shot, x_query


def make_nk_label(n, k, ep_per_batch=1):
    """[0]*k + [1]*k + ... + [n-1]*k, tiled per episode."""
    label = torch.arange(n).unsqueeze(1).expand(n, k).reshape(-1)
    return label.repeat(ep_per_batch)
