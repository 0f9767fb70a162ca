"""Online feature normaliser with the reference's API and state layout (src/migration/normalizer.py:9-75).

Statistics are plain tensor attributes (not buffers), exactly like the reference, so they are absent from
``state_dict`` and present in pickles.  The arithmetic is a handful of [rows, <=9] element-wise torch ops on the
device -- it is not inside the GNN and not a kernel worth owning; ``all_reduce_stats`` is the data-parallel hook.
"""
import torch
from torch import nn, Tensor

from .util import device


class Normalizer(nn.Module):
    def __init__(self, size: int, name: str, max_accumulations=10 ** 6, std_epsilon=1e-8) -> None:
        super().__init__()
        self._name = name
        self._max_accumulations = max_accumulations
        self._std_epsilon = torch.tensor([std_epsilon], requires_grad=False).to(device)
        self._acc_count = torch.zeros(1, dtype=torch.float32, requires_grad=False).to(device)
        self._num_accumulations = torch.zeros(1, dtype=torch.float32, requires_grad=False).to(device)
        self._acc_sum = torch.zeros(size, dtype=torch.float32, requires_grad=False).to(device)
        self._acc_sum_squared = torch.zeros(size, dtype=torch.float32, requires_grad=False).to(device)
        self._host_num_acc = 0              # host mirror of _num_accumulations: no device sync per call

    def forward(self, batched_data: Tensor, accumulate=True) -> Tensor:
        if accumulate and self._host_num_acc < self._max_accumulations:
            self._accumulate(batched_data)
        return (batched_data - self._mean()) / self._std_with_epsilon()

    def inverse(self, normalized_batch_data: Tensor) -> Tensor:
        return normalized_batch_data * self._std_with_epsilon() + self._mean()

    def _accumulate(self, batched_data: Tensor, reduce_fn=None) -> None:
        count = torch.tensor([float(batched_data.shape[0])], dtype=torch.float32, device=batched_data.device)
        data_sum = torch.sum(batched_data, dim=0)
        squared_data_sum = torch.sum(batched_data ** 2, dim=0)
        hook = reduce_fn or getattr(self, '_reduce_fn', None)
        if hook is not None:                 # data-parallel: every rank accumulates the GLOBAL batch statistics
            count, data_sum, squared_data_sum = hook(count, data_sum, squared_data_sum)
        self._acc_sum = self._acc_sum.add(data_sum)
        self._acc_sum_squared = self._acc_sum_squared.add(squared_data_sum)
        self._acc_count = self._acc_count.add(count)
        self._num_accumulations = self._num_accumulations.add(1.)
        self._host_num_acc += 1

    def _mean(self) -> Tensor:
        safe_count = torch.clamp(self._acc_count, min=1.0)
        return self._acc_sum / safe_count

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
