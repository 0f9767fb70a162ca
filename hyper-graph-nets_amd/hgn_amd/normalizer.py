"""Online feature normaliser with the reference's API and state layout (src/migration/normalizer.py:9-75).

Statistics are plain tensor attributes (not buffers), exactly like the reference, so they are absent from
``state_dict`` and present in pickles.  The arithmetic runs in three small HIP launches (include/hgn_features.h):
column statistics of the batch (fp64 accumulation, fixed order), the running-statistics update (gated on the device),
and the element-wise normalisation; ``_reduce_fn`` is the data-parallel hook between the first two
(parallel.attach_normalizer_sync: every rank accumulates the statistics of the GLOBAL batch).
"""
import torch
from torch import nn, Tensor

from . import features
from .util import device


class Normalizer(nn.Module):
    def __init__(self, size: int, name: str, max_accumulations=10 ** 6, std_epsilon=1e-8) -> None:
        super().__init__()
        self._name = name
        self._size = size
        self._max_accumulations = max_accumulations
        self._eps = float(std_epsilon)
        self._std_epsilon = torch.tensor([std_epsilon], requires_grad=False).to(device)
        self._acc_count = torch.zeros(1, dtype=torch.float32, requires_grad=False).to(device)
        self._num_accumulations = torch.zeros(1, dtype=torch.float32, requires_grad=False).to(device)
        self._acc_sum = torch.zeros(size, dtype=torch.float32, requires_grad=False).to(device)
        self._acc_sum_squared = torch.zeros(size, dtype=torch.float32, requires_grad=False).to(device)
        self._host_num_acc = 0              # host mirror of _num_accumulations: no device sync per call

    def forward(self, batched_data: Tensor, accumulate=True) -> Tensor:
        if accumulate and self._host_num_acc < self._max_accumulations:
            self._accumulate(batched_data)
        return features.normalize(batched_data, self._acc_sum, self._acc_sum_squared, self._acc_count, self._eps)

    def inverse(self, normalized_batch_data: Tensor) -> Tensor:
        return features.normalize(normalized_batch_data, self._acc_sum, self._acc_sum_squared, self._acc_count,
                                  self._eps, inverse=True)

    def _accumulate(self, batched_data: Tensor, reduce_fn=None) -> None:
        batch = features.col_stats(batched_data)                                  # [2F]: sums, sums of squares
        count = torch.full((1,), float(batched_data.shape[0]), dtype=torch.float32, device=batch.device)
        hook = reduce_fn or getattr(self, '_reduce_fn', None)
        if hook is not None:                 # data-parallel: every rank accumulates the GLOBAL batch statistics
            F = self._size
            count, s, q = hook(count, batch[:F], batch[F:])
            batch = torch.cat([s, q])
        features.normalizer_update(self._acc_sum, self._acc_sum_squared, self._acc_count, self._num_accumulations,
                                   batch, count, self._max_accumulations)
        self._host_num_acc += 1

    def _mean(self) -> Tensor:
        return self._acc_sum / torch.clamp(self._acc_count, min=1.0)

    def _std_with_epsilon(self) -> Tensor:
        safe_count = torch.clamp(self._acc_count, min=1.0)
        std = torch.sqrt(torch.abs(self._acc_sum_squared / safe_count - self._mean() ** 2))
        return torch.maximum(std, self._std_epsilon)

    def get_acc_sum(self) -> Tensor:
        return self._acc_sum

    def __setstate__(self, state):
        super().__setstate__(state)
        if '_host_num_acc' not in self.__dict__:
            self._host_num_acc = int(self._num_accumulations.item())
        if '_eps' not in self.__dict__:
            self._eps = float(self._std_epsilon.item())
        if '_size' not in self.__dict__:
            self._size = int(self._acc_sum.numel())
