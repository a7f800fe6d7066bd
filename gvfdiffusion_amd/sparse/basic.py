"""SparseTensor: batch of variable-length voxel feature lists, with the reference's container surface
(sparse/basic.py:18-463) but no spconv / torchsparse object underneath -- on this path the container only
carries `feats [sum L, ...]`, `coords [sum L, 4] (batch, x, y, z) int32`, batch-contiguous `layout` slices and
a scale-keyed spatial cache (window partitions, serialisations) for the attention operators."""
from typing import *

import torch

__all__ = ["SparseTensor", "sparse_batch_broadcast", "sparse_batch_op", "sparse_cat", "sparse_unbind"]


class SparseTensor:
    def __init__(self, feats: torch.Tensor, coords: torch.Tensor, shape: Optional[torch.Size] = None,
                 layout: Optional[List[slice]] = None, scale: Tuple[int, int, int] = (1, 1, 1),
                 spatial_cache: Optional[dict] = None):
        assert feats.shape[0] == coords.shape[0], f"Invalid feats shape: {feats.shape}, coords shape: {coords.shape}"
        assert coords.dim() == 2 and coords.shape[1] == 4, "coords must be [N, 4] (batch, x, y, z)"
        self.feats = feats
        self.coords = coords.int() if coords.dtype != torch.int32 else coords
        self._shape = shape
        self._layout = layout
        self._scale = tuple(scale)
        self._spatial_cache = spatial_cache if spatial_cache is not None else {}

    # ---- structure ------------------------------------------------------------------------------------
    def _batch_size(self) -> int:
        return int(self.coords[:, 0].max().item()) + 1 if self.coords.shape[0] > 0 else 0

    @property
    def shape(self) -> torch.Size:
        if self._shape is None:
            self._shape = torch.Size([self._batch_size(), *self.feats.shape[1:]])
        return self._shape

    def dim(self) -> int:
        return len(self.shape)

    @property
    def layout(self) -> List[slice]:
        if self._layout is None:
            B = self.shape[0]
            counts = torch.bincount(self.coords[:, 0].long(), minlength=B)
            ends = torch.cumsum(counts, 0).tolist()
            starts = [0] + ends[:-1]
            self._layout = [slice(s, e) for s, e in zip(starts, ends)]
            b = self.coords[:, 0]
            assert bool((b[1:] >= b[:-1]).all()), "SparseTensor: batch indices must be contiguous (sorted)"
        return self._layout

    @property
    def dtype(self):
        return self.feats.dtype

    @property
    def device(self):
        return self.feats.device

    # ---- conversions ------------------------------------------------------------------------------------
    def replace(self, feats: torch.Tensor, coords: Optional[torch.Tensor] = None) -> "SparseTensor":
        same = coords is None
        return SparseTensor(feats, self.coords if same else coords,
                            shape=torch.Size([self.shape[0], *feats.shape[1:]]) if same else None,
                            layout=self._layout if same else None, scale=self._scale,
                            spatial_cache=self._spatial_cache if same else None)

    def to(self, *args, **kwargs) -> "SparseTensor":
        device = kwargs.get("device")
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
            else:
                device = a
        feats = self.feats.to(device=device, dtype=dtype)
        coords = self.coords.to(device=device)
        return SparseTensor(feats, coords, self._shape, self._layout, self._scale,
                            self._spatial_cache if device is None else None)

    def type(self, dtype):
        return self.replace(self.feats.type(dtype))

    def cpu(self):
        return self.to("cpu")

    def cuda(self):
        return self.to("cuda")

    def half(self):
        return self.replace(self.feats.half())

    def float(self):
        return self.replace(self.feats.float())

    def detach(self):
        return self.replace(self.feats.detach())

    def reshape(self, *shape) -> "SparseTensor":
        return self.replace(self.feats.reshape(self.feats.shape[0], *shape))

    def unbind(self, dim: int) -> List["SparseTensor"]:
        return sparse_unbind(self, dim)

    def dense(self) -> torch.Tensor:
        """(B, C..., X, Y, Z) dense grid, zero where no voxel is active."""
        B = self.shape[0]
        ext = (self.coords[:, 1:].max(dim=0).values + 1).tolist() if self.coords.shape[0] else [0, 0, 0]
        out = torch.zeros((B, *ext, *self.feats.shape[1:]), dtype=self.feats.dtype, device=self.device)
        c = self.coords.long()
        out[c[:, 0], c[:, 1], c[:, 2], c[:, 3]] = self.feats
        nd = self.feats.dim() - 1
        return out.permute(0, *range(4, 4 + nd), 1, 2, 3)

    @staticmethod
    def full(aabb, dim, value, dtype=torch.float32, device=None) -> "SparseTensor":
        N, C = dim
        x = torch.arange(aabb[0], aabb[3] + 1)
        y = torch.arange(aabb[1], aabb[4] + 1)
        z = torch.arange(aabb[2], aabb[5] + 1)
        grid = torch.stack(torch.meshgrid(x, y, z, indexing="ij"), dim=-1).reshape(-1, 3)
        coords = torch.cat([torch.arange(N).view(-1, 1).repeat(1, grid.shape[0]).view(-1, 1), grid.repeat(N, 1)], dim=1)
        feats = torch.full((coords.shape[0], C), value, dtype=dtype)
        return SparseTensor(feats.to(device), coords.int().to(device))

    # ---- arithmetic ---------------------------------------------------------------------------------------
    def __elemwise__(self, other, op) -> "SparseTensor":
        if isinstance(other, torch.Tensor):
            try:
                other = torch.broadcast_to(other, self.shape)
                other = sparse_batch_broadcast(self, other)
            except RuntimeError:
                pass
        if isinstance(other, SparseTensor):
            other = other.feats
        return self.replace(op(self.feats, other))

    def __neg__(self):
        return self.replace(-self.feats)

    def __add__(self, o):
        return self.__elemwise__(o, torch.add)

    __radd__ = __add__

    def __sub__(self, o):
        return self.__elemwise__(o, torch.sub)

    def __rsub__(self, o):
        return self.__elemwise__(o, lambda a, b: torch.sub(b, a))

    def __mul__(self, o):
        return self.__elemwise__(o, torch.mul)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self.__elemwise__(o, torch.div)

    def __rtruediv__(self, o):
        return self.__elemwise__(o, lambda a, b: torch.div(b, a))

    def __getitem__(self, idx):
        if isinstance(idx, int):
            idx = [idx]
        elif isinstance(idx, slice):
            idx = range(*idx.indices(self.shape[0]))
        elif isinstance(idx, torch.Tensor):
            if idx.dtype == torch.bool:
                assert idx.shape == (self.shape[0],), f"Invalid index shape: {idx.shape}"
                idx = idx.nonzero().squeeze(1)
            idx = idx.tolist()
        coords, feats = [], []
        for new_idx, old_idx in enumerate(idx):
            sl = self.layout[old_idx]
            c = self.coords[sl].clone()
            c[:, 0] = new_idx
            coords.append(c)
            feats.append(self.feats[sl])
        return SparseTensor(torch.cat(feats, dim=0).contiguous(), torch.cat(coords, dim=0).contiguous())

    # ---- spatial cache ----------------------------------------------------------------------------------------
    def register_spatial_cache(self, key, value) -> None:
        self._spatial_cache.setdefault(str(self._scale), {})[key] = value

    def get_spatial_cache(self, key=None):
        cur = self._spatial_cache.get(str(self._scale), {})
        return cur if key is None else cur.get(key, None)


def sparse_batch_broadcast(input: SparseTensor, other: torch.Tensor) -> torch.Tensor:
    """(B, ...) per-sample tensor -> (sum L, ...) rows aligned with input.feats."""
    out = torch.empty((input.feats.shape[0], *other.shape[1:]), dtype=other.dtype, device=input.feats.device)
    for k, sl in enumerate(input.layout):
        out[sl] = other[k]
    return out


def sparse_batch_op(input: SparseTensor, other: torch.Tensor, op: callable = torch.add) -> SparseTensor:
    return input.replace(op(input.feats, sparse_batch_broadcast(input, other)))


def sparse_cat(inputs: List[SparseTensor], dim: int = 0) -> SparseTensor:
    if dim == 0:
        start, coords = 0, []
        for t in inputs:
            c = t.coords.clone()
            c[:, 0] += start
            coords.append(c)
            start += t.shape[0]
        return SparseTensor(torch.cat([t.feats for t in inputs], dim=0), torch.cat(coords, dim=0))
    return inputs[0].replace(torch.cat([t.feats for t in inputs], dim=dim))


def sparse_unbind(input: SparseTensor, dim: int) -> List[SparseTensor]:
    if dim == 0:
        return [input[i] for i in range(input.shape[0])]
    return [input.replace(f) for f in input.feats.unbind(dim)]
