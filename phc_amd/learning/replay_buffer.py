"""P9: AMP observation ring buffers (same semantics as the reference phc/learning/replay_buffer.py:3-84:
wrap-around store, sampling through a pre-shuffled index list, `% head` while the buffer is not yet full).
Storage lives in HBM (200k x 1960 fp32 = 1.57 GB per buffer at the shipped sizes)."""
import torch


class ReplayBuffer:
    def __init__(self, buffer_size, device):
        self._head = 0
        self._total_count = 0
        self._buffer_size = buffer_size
        self._device = device
        self._data_buf = None
        self._sample_idx = torch.randperm(buffer_size, device=device)
        self._sample_head = 0

    def reset(self):
        self._head = 0
        self._total_count = 0
        self._reset_sample_idx()

    def get_buffer_size(self):
        return self._buffer_size

    def get_total_count(self):
        return self._total_count

    def store(self, data_dict):
        if self._data_buf is None:
            self._data_buf = {k: torch.zeros((self._buffer_size,) + v.shape[1:], device=self._device, dtype=v.dtype) for k, v in data_dict.items()}
        n = next(iter(data_dict.values())).shape[0]
        size = self._buffer_size
        assert n <= size
        for key, buf in self._data_buf.items():
            d = data_dict[key]
            assert d.shape[0] == n
            store_n = min(n, size - self._head)
            buf[self._head:self._head + store_n] = d[:store_n]
            rem = n - store_n
            if rem > 0:
                buf[0:rem] = d[store_n:]
        self._head = (self._head + n) % size
        self._total_count += n

    def sample(self, n):
        size = self._buffer_size
        idx = torch.arange(self._sample_head, self._sample_head + n, device=self._device) % size
        rand_idx = self._sample_idx[idx]
        if self._total_count < size:
            rand_idx = rand_idx % self._head
        out = {k: v[rand_idx] for k, v in self._data_buf.items()}
        self._sample_head += n
        if self._sample_head >= size:
            self._reset_sample_idx()
        return out

    def _reset_sample_idx(self):
        self._sample_idx[:] = torch.randperm(self._buffer_size, device=self._device)
        self._sample_head = 0
