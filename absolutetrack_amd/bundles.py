"""Generic helpers over dataclasses / NamedTuples / dicts of tensors with the names the reference's callers
use (lib/data_utils/bundles.py: to_device :171, collate :178, map_fields :138, group :227, asdict :37)."""
import dataclasses
from typing import Any, Mapping

import numpy as np
import torch


def is_dictlike(obj: Any) -> bool:
    return dataclasses.is_dataclass(obj) or hasattr(obj, "_asdict") or isinstance(obj, Mapping)


def asdict(obj):
    if dataclasses.is_dataclass(obj):
        return {f.name: getattr(obj, f.name) for f in dataclasses.fields(obj)}
    if hasattr(obj, "_asdict"):
        return obj._asdict()
    if isinstance(obj, Mapping):
        return obj
    raise TypeError("asdict() requires a Mapping, dataclass, or NamedTuple")


def _rebuild(proto, items: dict):
    if isinstance(proto, Mapping):
        return type(proto)(items.items())
    return type(proto)(**items)


def map_fields(func, obj, only_type=object):
    """Apply func to every leaf (of type only_type) of a nested container, keeping the container types."""
    if is_dictlike(obj):
        return _rebuild(obj, {k: map_fields(func, v, only_type) for k, v in asdict(obj).items()})
    if isinstance(obj, tuple):
        return tuple(map_fields(func, v, only_type) for v in obj)
    if isinstance(obj, list):
        return [map_fields(func, v, only_type) for v in obj]
    return func(obj) if isinstance(obj, only_type) else obj


def to_device(obj, device):
    return map_fields(lambda t: t.to(device), obj, only_type=torch.Tensor)


def group(batch, group_fn):
    """Turn a list of N like-structured items into one item whose leaves are group_fn(list of N leaves)."""
    first = batch[0]
    if isinstance(first, (np.ndarray, np.generic, torch.Tensor)):
        return group_fn(batch)
    if is_dictlike(first):
        rows = [asdict(x) for x in batch]
        return _rebuild(first, {k: group([r[k] for r in rows], group_fn) for k in rows[0]})
    if isinstance(first, tuple):
        return tuple(group([b[i] for b in batch], group_fn) for i in range(len(first)))
    if isinstance(first, list):
        return [group([b[i] for b in batch], group_fn) for i in range(len(first))]
    if any(b is None for b in batch):
        if all(b is None for b in batch):
            return None
        raise TypeError("Some items are None and others are not")
    return batch


def collate(batch, device=None):
    """Stack arrays / tensors of N items along a new leading dimension."""
    def stack(leaves):
        if isinstance(leaves[0], (np.ndarray, np.generic)):
            return np.stack(leaves)
        if isinstance(leaves[0], torch.Tensor):
            t = torch.stack(leaves)
            return t if device is None else t.to(device)
        raise TypeError(f"Can't stack tensors: unknown type {type(leaves[0])} found")
    return group(batch, stack)
