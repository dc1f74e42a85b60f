"""Named, nested, repeat-indexed vectors ("power indexing").

The reference keeps every vector on the hot path (opt_x, opt_p, bounds, scalings,
multipliers) in casadi.tools structs and addresses them as e.g.
`opt_x_num['_u', 0, 0]`, `lb_opt_x['_x', 1:N, :, -1]`, `p_template['_p', :, names]`
(/root/reference/do_mpc/controller/_mpc.py:1024-1026, 1064-1075, 871).  This module
gives the same addressing over a flat float64 buffer with the *same canonical
order* (entries in declaration order, repeats row-major, matrices column-major),
because that order is what the C-ABI solver and the golden vectors use.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import sym


class Entry:
    """One field: a (n,m) matrix, or a sub-layout, optionally repeated over leading dims."""

    def __init__(self, name: str, shape: Union[int, Tuple[int, int]] = (1, 1),
                 struct: Optional["Layout"] = None, repeat: Union[int, Sequence[int], None] = None):
        self.name = name
        if isinstance(shape, int):
            shape = (shape, 1)
        self.shape = tuple(shape)
        self.struct = struct
        if repeat is None:
            repeat = []
        elif isinstance(repeat, (int, np.integer)):
            repeat = [int(repeat)]
        self.repeat = [int(r) for r in repeat]
        self.elem = struct.size if struct is not None else self.shape[0] * self.shape[1]
        self.size = int(np.prod(self.repeat, dtype=int)) * self.elem if self.repeat else self.elem


def entry(name, shape=(1, 1), struct=None, repeat=None) -> Entry:
    return Entry(name, shape, struct, repeat)


class Layout:
    """Ordered collection of entries -> flat offsets."""

    def __init__(self, entries: Sequence[Entry]):
        self.entries: List[Entry] = list(entries)
        self.by_name: Dict[str, Entry] = {e.name: e for e in self.entries}
        self.offsets: Dict[str, int] = {}
        off = 0
        for e in self.entries:
            self.offsets[e.name] = off
            off += e.size
        self.size = off

    def keys(self):
        return [e.name for e in self.entries]

    def labels(self) -> List[str]:
        out = []
        for e in self.entries:
            reps = list(np.ndindex(*e.repeat)) if e.repeat else [()]
            for r in reps:
                pre = "[" + ",".join([e.name] + [str(i) for i in r])
                if e.struct is not None:
                    for lab in e.struct.labels():
                        out.append(pre + "," + lab[1:])
                else:
                    for i in range(e.elem):
                        out.append(pre + f",{i}]")
        return out

    # -- index resolution -------------------------------------------------------
    def resolve(self, key) -> np.ndarray:
        """Flat indices selected by a power index.  Result keeps the selection's shape
        (repeat dims that were sliced, then the element dim) so values broadcast naturally."""
        if not isinstance(key, tuple):
            key = (key,)
        if len(key) == 0:
            return np.arange(self.size)
        name = key[0]
        rest = key[1:]
        if isinstance(name, (list, tuple)):
            parts = [self.resolve((n,) + rest) for n in name]
            return np.concatenate([p.reshape(p.shape[:-1] + (-1,)) if p.ndim else p.reshape(1) for p in parts], axis=-1)
        if isinstance(name, slice):
            assert name == slice(None)
            return self.resolve((self.keys(),) + rest)
        if name not in self.by_name:
            raise KeyError(f"'{name}' not in {self.keys()}")
        e = self.by_name[name]
        base = self.offsets[name]
        nrep = len(e.repeat)
        rep_keys = list(rest[:nrep])
        rest = rest[nrep:]
        while len(rep_keys) < nrep:
            rep_keys.append(slice(None))
        # repeat selection
        idx = np.zeros((), dtype=int) + base
        stride = e.elem
        strides = []
        for r in reversed(e.repeat):
            strides.append(stride)
            stride *= r
        strides = strides[::-1]
        for k, n, st in zip(rep_keys, e.repeat, strides):
            sel = np.arange(n)[k]
            if np.ndim(sel) == 0:
                idx = idx + int(sel) * st
            else:
                idx = idx[..., None] + sel * st
        # element selection
        if e.struct is not None:
            inner = e.struct.resolve(tuple(rest)) if rest else np.arange(e.struct.size)
        else:
            if rest:
                k = rest[0] if len(rest) == 1 else tuple(rest)
                lin = np.arange(e.elem).reshape(e.shape, order="F")
                inner = np.atleast_1d(lin[k] if isinstance(k, tuple) else lin.reshape(-1, order="F")[k]).reshape(-1, order="F")
            else:
                inner = np.arange(e.elem)
        return idx[..., None] + inner


class NumStruct:
    """Flat float64 vector addressed through a Layout (casadi DMStruct analogue)."""

    def __init__(self, layout: Layout, value: Union[float, np.ndarray] = 0.0):
        self.layout = layout
        if np.isscalar(value):
            self.master = np.full(layout.size, float(value))
        else:
            v = np.asarray(value.arr if isinstance(value, sym.DM) else value, dtype=float).reshape(-1, order="F")
            assert v.size == layout.size, f"size {v.size} != {layout.size}"
            self.master = v.copy()

    # casadi-isms
    @property
    def cat(self) -> sym.DM:
        return sym.DM(self.master.reshape(-1, 1))

    @property
    def shape(self):
        return (self.layout.size, 1)

    @property
    def size(self):
        return self.layout.size

    def keys(self):
        return self.layout.keys()

    def labels(self):
        return self.layout.labels()

    def full(self):
        return self.master.reshape(-1, 1).copy()

    def __array__(self, dtype=None, copy=None):
        return self.master if dtype is None else self.master.astype(dtype)

    def __call__(self, value=0.0) -> "NumStruct":
        return NumStruct(self.layout, value)

    def __getitem__(self, key):
        idx = self.layout.resolve(key)
        out = self.master[idx]
        if isinstance(key, tuple) and any(isinstance(k, slice) for k in key[1:1 + 8]):
            # sliced repeats -> list of column vectors like casadi
            if out.ndim >= 2:
                flat = out.reshape(-1, out.shape[-1])
                return [sym.DM(r.reshape(-1, 1)) for r in flat]
        e = self._leaf_entry(key)
        if e is not None and out.ndim == 1 and out.size == e.shape[0] * e.shape[1]:
            return sym.DM(out.reshape(e.shape, order="F"))
        return sym.DM(out.reshape(-1, 1))

    def _leaf_entry(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        lay = self.layout
        e = None
        for k in key:
            if isinstance(k, str) and k in lay.by_name:
                e = lay.by_name[k]
                if e.struct is not None:
                    lay = e.struct
                    e = None
        return e

    def __setitem__(self, key, value):
        idx = self.layout.resolve(key)
        if isinstance(value, sym.DM):
            value = value.arr
        elif isinstance(value, NumStruct):
            value = value.master
        elif isinstance(value, sym.SX):
            value = value.to_numpy()
        v = np.asarray(value, dtype=float)
        if v.ndim >= 2 and v.size == idx.size:
            # matrix-valued entry: column-major flattening; list-of-tuples (p_scenario) stays row-major
            if idx.ndim == 1:
                v = v.reshape(-1, order="F")
            else:
                v = v.reshape(idx.shape)
        elif v.size == idx.shape[-1] and idx.ndim > 1:
            v = v.reshape(-1)
        elif v.size == idx.size:
            v = v.reshape(idx.shape)
        elif idx.ndim == 2 and v.size == idx.shape[0]:
            v = v.reshape(-1, 1)          # a list of scalars over a sliced repeat: one value per block (casadi payload unpacking)
        self.master[idx] = v

    def get(self, *key) -> np.ndarray:
        """Raw ndarray view of a selection (shape: sliced repeats + element)."""
        return self.master[self.layout.resolve(tuple(key))]

    def __repr__(self):
        return f"NumStruct({self.layout.keys()}, size={self.layout.size})"


class SymStruct:
    """Symbolic counterpart: one SX vector + the same addressing (struct_symSX analogue)."""

    def __init__(self, layout: Layout, prefix: str = "v"):
        self.layout = layout
        nodes = [sym.symbol(f"{prefix}{lab}") for lab in layout.labels()] if layout.size < 4096 else \
                [sym.symbol(f"{prefix}_{i}") for i in range(layout.size)]
        self.vec = sym.SX(nodes, (layout.size, 1))

    @property
    def cat(self) -> sym.SX:
        return self.vec

    @property
    def shape(self):
        return (self.layout.size, 1)

    @property
    def size(self):
        return self.layout.size

    def keys(self):
        return self.layout.keys()

    def labels(self):
        return self.layout.labels()

    def __getitem__(self, key) -> sym.SX:
        idx = self.layout.resolve(key).reshape(-1)
        e = None
        if isinstance(key, str):
            e = self.layout.by_name.get(key)
        nodes = [self.vec.data[i] for i in idx]
        if e is not None and e.struct is None and not e.repeat:
            return sym.SX(nodes, e.shape)
        return sym.SX(nodes, (len(nodes), 1))

    def __call__(self, value=0.0) -> NumStruct:
        return NumStruct(self.layout, value)
