"""Row-aligned containers used at the API boundary of the hot path.

`TensorCollection` = named tensors sharing a leading row dimension; `PandasTensorCollection` additionally carries a
pandas frame (`infos`, index always 0..n-1) with one row per tensor row.  Behavioural contract taken from how the
reference's callers use them (src/megapose/utils/tensor_collection.py:44-197; callers in
src/megapose/inference/pose_estimator.py): attribute access to tensors, row selection with ints/lists/tensors,
`.infos`, `len()`, `.cuda()/.cpu()/.float()`, `clone()`, pickling, and `concatenate`.
Multi-rank gathering is done over RCCL in `megapose6d_amd.distributed` (the reference goes through files, :165-186).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Dict, Iterable

import pandas as pd
import torch

_STORE = "_tensors"


class TensorCollection:
    def __init__(self, **named: torch.Tensor):
        object.__setattr__(self, _STORE, OrderedDict())
        for name, value in named.items():
            self.register_tensor(name, value)

    # -- storage ---------------------------------------------------------------------------------
    @property
    def tensors(self) -> Dict[str, torch.Tensor]:
        return object.__getattribute__(self, _STORE)

    def register_tensor(self, name: str, tensor: torch.Tensor) -> None:
        self.tensors[name] = tensor

    def delete_tensor(self, name: str) -> None:
        self.tensors.pop(name)

    def __getattr__(self, name):  # only reached when normal lookup fails
        try:
            store = object.__getattribute__(self, _STORE)
        except AttributeError:
            raise AttributeError(name) from None
        if name in store:
            return store[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        try:
            store = object.__getattribute__(self, _STORE)
        except AttributeError:
            raise ValueError("Please call __init__") from None
        if name in store:
            store[name] = value
        else:
            object.__setattr__(self, name, value)

    # -- transforms ------------------------------------------------------------------------------
    def _map_inplace(self, fn: Callable[[torch.Tensor], torch.Tensor]):
        store = self.tensors
        for key in list(store):
            store[key] = fn(store[key])
        return self

    def to(self, target):
        return self._map_inplace(lambda t: t.to(target))

    def cuda(self):
        return self.to("cuda")

    def cpu(self):
        return self.to("cpu")

    def float(self):
        return self.to(torch.float)

    def double(self):
        return self.to(torch.double)

    def half(self):
        return self.to(torch.half)

    def _select(self, ids) -> Dict[str, torch.Tensor]:
        return {key: t[ids] for key, t in self.tensors.items()}

    def __getitem__(self, ids):
        return TensorCollection(**self._select(ids))

    def clone(self):
        return TensorCollection(**{key: t.clone() for key, t in self.tensors.items()})

    @property
    def device(self):
        return next(iter(self.tensors.values())).device

    def _describe(self) -> str:
        return "".join(f"    {key}: {t.shape} {t.dtype} {t.device},\n" for key, t in self.tensors.items())

    def __repr__(self):
        return f"{type(self).__name__}(\n{self._describe()})"

    def __getstate__(self):
        return {"tensors": dict(self.tensors)}

    def __setstate__(self, state):
        TensorCollection.__init__(self, **state["tensors"])


class PandasTensorCollection(TensorCollection):
    def __init__(self, infos: pd.DataFrame, **named: torch.Tensor):
        super().__init__(**named)
        self.infos = infos.reset_index(drop=True)
        self.meta = {}

    def __len__(self) -> int:
        return len(self.infos)

    def __getitem__(self, ids):
        rows = ids.cpu().numpy() if torch.is_tensor(ids) else ids
        return PandasTensorCollection(self.infos.iloc[rows].reset_index(drop=True), **self._select(ids))

    def clone(self):
        return PandasTensorCollection(self.infos.copy(), **{key: t.clone() for key, t in self.tensors.items()})

    def merge_df(self, df: pd.DataFrame, *args, **kwargs):
        merged = self.infos.merge(df, how="left", *args, **kwargs)
        if len(merged) != len(self.infos):
            raise AssertionError("merge_df must not change the number of rows")
        return PandasTensorCollection(merged, **self.tensors)

    def __repr__(self):
        return f"{type(self).__name__}(\n{self._describe()}{'-' * 40}\n    infos:\n{self.infos!r}\n)"

    def __getstate__(self):
        return {"tensors": dict(self.tensors), "infos": self.infos, "meta": self.meta}

    def __setstate__(self, state):
        PandasTensorCollection.__init__(self, state["infos"], **state["tensors"])
        self.meta = state["meta"]


def concatenate(parts: Iterable[PandasTensorCollection]) -> PandasTensorCollection:
    parts = [p for p in parts if len(p) > 0]
    if not parts:
        return PandasTensorCollection(infos=pd.DataFrame())
    frame = pd.concat([p.infos for p in parts], axis=0, sort=False).reset_index(drop=True)
    keys = list(parts[0].tensors)
    return PandasTensorCollection(frame, **{k: torch.cat([p.tensors[k] for p in parts], dim=0) for k in keys})
