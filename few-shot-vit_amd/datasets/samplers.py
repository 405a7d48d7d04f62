"""Episode index sampler (surface and RNG stream of test_phase/datasets/samplers.py:5-35)."""
import ctypes as C

import numpy as np
import torch


class CategoriesSampler:
    """Yields, per batch, a flat LongTensor of ep_per_batch * n_cls * n_per dataset indices in
    class-major order.  Draws from the GLOBAL legacy numpy RNG (`np.random.choice`) in the main
    process, exactly like the reference, so `np.random.seed(s)` reproduces its episode stream.

    Extensions (the defaults are the reference's behaviour):

    * `rank` / `world_size`: batches are sharded `rank::world_size` over the GPUs of a node without any data-path collective.
      `shard='replay'`: every rank draws the SAME global stream and keeps its batches.  `shard='scatter'`: rank 0 alone draws the
      stream - the whole index table of an epoch, n_batch x ep_per_batch x n_cls x n_per int64 (1.6 MB at the reference's 2000 x 100) -
      and ONE `torch.distributed.broadcast` hands it to the others, which do not touch their generators; the host work of a rank no
      longer grows with the number of ranks it does not serve.  'scatter' puts a blocking broadcast inside `__iter__`: EVERY rank must iterate
      the sampler the same number of times (a rank-0-only validation loop would hang), and the other ranks' generators do not advance - so it
      is opt-in (`test_few_shot.evaluate` / `train_meta` ask for it, their ranks run in lockstep); `shard=None` = 'replay'.
    * `native`: the draws run in libfsvit's `fsvit_sampler_draw` on the generator state taken from `np.random.get_state()` and handed
      back with `set_state()` - the same MT19937 outputs through the same rejection sampling and Fisher-Yates order as numpy's
      `choice(replace=False)`, bit-identical indices AND generator state afterwards, ~7 x less host time per episode
      (tests/test_sampler_native_cpu.py).  `native=None`: native when the library loads and the generator is MT19937, numpy when the native
      draw reports an error.  The native path draws AHEAD in slabs (64, 256, 1024 batches): the generator is where the reference's lazy per-batch
      draws would leave it only at slab boundaries and at the end of the epoch - a caller that breaks out early, or uses np.random between
      batches, passes `native_slab=1` (one batch per native call) or `native=False`."""

    def __init__(self, label, n_batch, n_cls, n_per, ep_per_batch=1, rank=0, world_size=1, shard=None, native=None, native_slab=None):
        self.n_batch = n_batch
        self.n_cls = n_cls
        self.n_per = n_per
        self.ep_per_batch = ep_per_batch
        self.rank, self.world_size = rank, world_size
        if shard not in (None, 'replay', 'scatter'):
            raise ValueError(shard)
        self.shard, self.native, self.native_slab = shard, native, native_slab
        label = np.array(label)
        self.catlocs = [np.argwhere(label == c).reshape(-1) for c in range(max(label) + 1)]
        self._items = np.ascontiguousarray(np.concatenate(self.catlocs).astype(np.int64))
        self._offsets = np.zeros(len(self.catlocs) + 1, dtype=np.int64)
        np.cumsum([len(c) for c in self.catlocs], out=self._offsets[1:])

    def __len__(self):
        return len(range(self.rank, self.n_batch, self.world_size))

    # ---- drawing
    def _native_lib(self):
        if self.native is False:
            return None
        try:
            from .. import _lib
            return _lib.load()
        except Exception:
            if self.native:
                raise
            return None

    def _draw_native(self, lib, n_batch):
        """n_batch batches from the global legacy generator through fsvit_sampler_draw -> int64 [n_batch, ep, n_cls, n_per], or None (not MT19937)."""
        st = np.random.get_state()
        if st[0] != 'MT19937':
            return None
        key = np.ascontiguousarray(st[1], dtype=np.uint32).copy()
        pos = C.c_int(int(st[2]))
        out = np.empty((n_batch, self.ep_per_batch, self.n_cls, self.n_per), dtype=np.int64)
        from .. import _lib
        _lib.check(lib.fsvit_sampler_draw(C.c_void_p(key.ctypes.data), C.cast(C.byref(pos), C.c_void_p), C.c_void_p(self._offsets.ctypes.data),
                                          C.c_void_p(self._items.ctypes.data), len(self.catlocs), n_batch, self.ep_per_batch, self.n_cls, self.n_per,
                                          C.c_void_p(out.ctypes.data)))
        np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
        return out

    def _draw_numpy(self, n_batch):
        # The legacy-RNG call sequence is the reference's (one choice of classes, then one choice per class, per episode); the indices are
        # assembled in one numpy array per batch (the reference's per-class torch.from_numpy + two torch.stack cost more host time than the
        # draws themselves).
        n_cat, choice = len(self.catlocs), np.random.choice
        for _ in range(n_batch):
            batch = np.empty((self.ep_per_batch, self.n_cls, self.n_per), dtype=np.int64)
            for e in range(self.ep_per_batch):
                classes = choice(n_cat, self.n_cls, replace=False)
                for j, c in enumerate(classes):
                    batch[e, j] = choice(self.catlocs[c], self.n_per, replace=False)
            yield batch

    def _stream(self):
        """Every batch of the epoch in global stream order, drawn on THIS process's generator."""
        lib = self._native_lib()
        if lib is not None:
            # in slabs, so that the first batches are out while a long epoch is still being drawn (the consumer launches GPU work in between)
            done, slab = 0, self.native_slab or 64
            while done < self.n_batch:
                n = min(slab, self.n_batch - done)
                try:
                    tab = self._draw_native(lib, n)
                except Exception:
                    if self.native:
                        raise
                    tab = None                               # (e.g. a class shorter than n_per: numpy raises only when that class is drawn)
                if tab is None:
                    break
                for b in tab:
                    yield b
                done += n
                if not self.native_slab:
                    slab = min(4 * slab, 1024)
            if done == self.n_batch:
                return
            remaining = self.n_batch - done
        else:
            remaining = self.n_batch
        yield from self._draw_numpy(remaining)

    def _shard_mode(self):
        if self.world_size == 1:
            return 'replay'
        return self.shard or 'replay'

    def __iter__(self):
        if self._shard_mode() == 'scatter':
            import torch.distributed as dist
            shape = (self.n_batch, self.ep_per_batch * self.n_cls * self.n_per)
            if self.rank == 0:
                table = torch.from_numpy(np.stack(list(self._stream())).reshape(shape)) if self.n_batch else torch.zeros(shape, dtype=torch.int64)
            else:
                table = torch.empty(shape, dtype=torch.int64)
            if dist.get_backend() == 'nccl':              # RCCL moves device buffers: 1.6 MB each way, once per epoch
                dev = torch.device('cuda', torch.cuda.current_device())
                t = table.to(dev)
                dist.broadcast(t, src=0)
                table = t.cpu()
            else:
                dist.broadcast(table, src=0)
            for i_batch in range(self.rank, self.n_batch, self.world_size):
                yield table[i_batch]
            return
        for i_batch, batch in enumerate(self._stream()):
            if i_batch % self.world_size != self.rank:
                continue                      # drawn (keeps the stream aligned) but owned by another rank
            yield torch.from_numpy(np.ascontiguousarray(batch)).view(-1)  # bs * n_cls * n_per
