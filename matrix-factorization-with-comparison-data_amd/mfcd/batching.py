"""Host-side batch stream: turn the caller's DataLoader into what the device
path consumes — one packed record array per dataset and one index order per epoch — while drawing
from torch's RNG streams exactly as iterating the DataLoader would, so that everything downstream
of the global generator (e.g. the randperm at structure.py:390) stays aligned with the reference.

Reference behaviour restated here (not copied): structure.py:738-740 builds
DataLoader(dataset, batch_size=64, shuffle=True/False); iterating one draws, in this order,
  1. a base seed: one int64 from loader.generator (None = the global generator),
  2. for shuffle=True: RandomSampler draws one int64 seed from the global generator, seeds a private
     generator with it and takes torch.randperm(N) from that generator.
The last batch is short (drop_last=False by default) and is not dropped.
"""
import numpy as np
import torch
from torch.utils.data import RandomSampler, SequentialSampler


def dataset_records(dataset):
    """(u, i, j, z) rows of a dataset as a float64 array [N, 4].

    Fast path: the reference's BTLPreferenceDataset keeps a Python list of 4-tuples in `.data`
    (structure.py:491, 527-531).  Anything else is read item by item."""
    fn = getattr(dataset, "_mfcd_records", None)   # this build's BTLPreferenceDataset: rows already an array
    rows = fn() if callable(fn) else None
    if rows is not None:
        return np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 4)
    data = getattr(dataset, "data", None)
    if isinstance(data, np.ndarray) and data.ndim == 2 and data.shape[1] == 4:
        return np.ascontiguousarray(data, dtype=np.float64)
    if isinstance(data, (list, tuple)):
        return np.asarray(data, dtype=np.float64).reshape(-1, 4)
    return np.asarray([tuple(float(x) for x in dataset[k]) for k in range(len(dataset))],
                      dtype=np.float64).reshape(-1, 4)


def pack_records(rows, n=None, m=None):
    """float64 [N,4] rows -> int32 [N,4] array in the 16-byte mfcd_sample layout (z as fp32 bits).

    Raises IndexError for indices outside the tables, as U[u] / V[i] would (structure.py:787-789)."""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 4)
    rec = np.empty((rows.shape[0], 4), dtype=np.int32)
    idx = rows[:, :3]
    if rows.shape[0]:
        if n is not None and ((idx[:, 0] < -n).any() or (idx[:, 0] >= n).any()):
            raise IndexError("user index out of range for U")
        if m is not None and ((idx[:, 1:] < -m).any() or (idx[:, 1:] >= m).any()):
            raise IndexError("item index out of range for V")
    iu = idx.astype(np.int64)
    if n is not None:
        iu[:, 0] = np.where(iu[:, 0] < 0, iu[:, 0] + n, iu[:, 0])  # Python-style negative indexing
    if m is not None:
        iu[:, 1:] = np.where(iu[:, 1:] < 0, iu[:, 1:] + m, iu[:, 1:])
    rec[:, :3] = iu
    rec[:, 3] = rows[:, 3].astype(np.float32).view(np.int32)  # z.float() of structure.py:849
    return rec


def _draw_base_seed(loader):
    torch.empty((), dtype=torch.int64).random_(generator=loader.generator)


def epoch_order(loader):
    """Index order one pass over `loader` would visit, as an int64 tensor, consuming RNG identically.

    Returns (order, batch_size).  `order` already reflects drop_last."""
    n_items = len(loader.dataset)
    bs = loader.batch_size
    if bs is None or loader.batch_sampler is None:
        raise NotImplementedError("the HIP path needs an auto-batching DataLoader (batch_size set)")
    _draw_base_seed(loader)
    sampler = loader.sampler
    if type(sampler) is SequentialSampler:
        order = torch.arange(n_items, dtype=torch.int64)
    elif (type(sampler) is RandomSampler and not sampler.replacement and sampler.num_samples == n_items):
        if sampler.generator is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            gen = torch.Generator()
            gen.manual_seed(seed)
        else:
            gen = sampler.generator
        order = torch.randperm(n_items, generator=gen)
    else:  # any other sampler: let it speak for itself
        order = torch.tensor([k for batch in loader.batch_sampler for k in batch], dtype=torch.int64)
        return order, bs
    if loader.drop_last:
        order = order[: (n_items // bs) * bs]
    return order, bs


def n_batches(n_items, bs):
    return (n_items + bs - 1) // bs
