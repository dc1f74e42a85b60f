"""Scalar expression DAG with automatic differentiation and C/HIP code generation.

This is the symbolic layer that replaces CasADi's SX on the hot path
(`casadi.SX`, `casadi.Function`, `casadi.jacobian` as used by
/root/reference/do_mpc/model/_model.py:937-1058 and
/root/reference/do_mpc/optimizer.py:789-996).  It is *not* a CasADi clone: it is
a hash-consed scalar DAG (common sub-expressions are shared by construction)
whose only consumers are

  * reverse / forward mode AD producing new DAG nodes (gradients, Jacobian
    blocks, Hessian-of-Lagrangian blocks of the per-collocation-point model
    functions), and
  * a straight-line code emitter that lowers a set of output expressions to a
    `__host__ __device__` function body for the gfx950 IPM kernels
    (do_mpc_amd/lowering.py), plus a NumPy emitter used by host-side
    bookkeeping (aux expressions).

Matrices follow CasADi's conventions where the reference relies on them:
column-major flattening, `vertcat`/`horzcat`, elementwise `*`, `@` for matmul.
"""
from __future__ import annotations

import math
import struct
import zlib
from typing import Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

# --------------------------------------------------------------------------- nodes

_UNARY = ("neg", "sq", "sqrt", "exp", "log", "sin", "cos", "tan", "tanh", "fabs",
          "sinh", "cosh", "asin", "acos", "atan", "sign", "inv")
_BINARY = ("add", "sub", "mul", "div", "pow", "fmin", "fmax", "atan2")


class Node:
    """One scalar operation.  Never construct directly: use the builders below."""
    __slots__ = ("op", "a", "b", "val", "idx", "skey")

    def __init__(self, op, a, b, val, idx):
        self.op = op
        self.a = a
        self.b = b
        self.val = val
        self.idx = idx          # identity (creation order, process-local)
        self.skey = _skey(op, a, b, val)   # structural key: same expression -> same key in any process

    def is_const(self):
        return self.op == "const"

    def __repr__(self):
        if self.op == "const":
            return repr(self.val)
        if self.op == "sym":
            return str(self.val)
        if self.b is None:
            return f"{self.op}({self.a!r})"
        return f"{self.op}({self.a!r},{self.b!r})"


_M64 = (1 << 64) - 1


def _skey(op, a, b, val) -> int:
    """Deterministic structural hash; used to order commutative operands so that generated code
    (and therefore the model hash / code-object cache key) does not depend on build history."""
    if op == "const":
        h = zlib.crc32(struct.pack("<d", val))
    elif op == "sym":
        h = zlib.crc32(str(val).encode())
    else:
        h = zlib.crc32(op.encode())
    h = (h * 0x9E3779B97F4A7C15 + 0x7F4A7C15) & _M64
    if a is not None:
        h = ((h ^ a.skey) * 0x100000001B3 + 1) & _M64
    if b is not None:
        h = ((h ^ (b.skey * 3 + 7)) * 0x100000001B3 + 2) & _M64
    return h


def _before(a: "Node", b: "Node") -> bool:
    return (a.skey, a.idx) <= (b.skey, b.idx)


_TABLE: Dict[tuple, Node] = {}
_COUNTER = [0]


def _mk(op, a=None, b=None, val=None) -> Node:
    if op == "sym":
        _COUNTER[0] += 1
        return Node(op, None, None, val, _COUNTER[0])  # symbols are unique objects
    key = (op, a.idx if a is not None else None, b.idx if b is not None else None, val)
    n = _TABLE.get(key)
    if n is None:
        _COUNTER[0] += 1
        n = Node(op, a, b, val, _COUNTER[0])
        _TABLE[key] = n
    return n


def const(v: float) -> Node:
    v = float(v)
    if v == 0.0:
        v = 0.0  # merge -0.0
    return _mk("const", val=v)


ZERO = const(0.0)
ONE = const(1.0)
MONE = const(-1.0)
TWO = const(2.0)


def symbol(name: str) -> Node:
    return _mk("sym", val=name)


def _is(n: Node, v: float) -> bool:
    return n.op == "const" and n.val == v


_PYFUN = {
    "neg": lambda a: -a, "sq": lambda a: a * a, "sqrt": math.sqrt, "exp": math.exp,
    "log": math.log, "sin": math.sin, "cos": math.cos, "tan": math.tan, "tanh": math.tanh,
    "fabs": abs, "sinh": math.sinh, "cosh": math.cosh, "asin": math.asin, "acos": math.acos,
    "atan": math.atan, "sign": lambda a: (a > 0) - (a < 0), "inv": lambda a: 1.0 / a,
}


def unary(op: str, a: Node) -> Node:
    if a.op == "const":
        try:
            return const(_PYFUN[op](a.val))
        except (ValueError, ZeroDivisionError, OverflowError):
            pass
    if op == "neg":
        if a.op == "neg":
            return a.a
        if a.op == "sub":
            return sub(a.b, a.a)
    return _mk(op, a)


def neg(a):
    return unary("neg", a)


def add(a: Node, b: Node) -> Node:
    if a.op == "const" and b.op == "const":
        return const(a.val + b.val)
    if _is(a, 0.0):
        return b
    if _is(b, 0.0):
        return a
    if b.op == "neg":
        return sub(a, b.a)
    if a.op == "neg":
        return sub(b, a.a)
    if not _before(a, b):  # canonical order for commutative ops
        a, b = b, a
    return _mk("add", a, b)


def sub(a: Node, b: Node) -> Node:
    if a.op == "const" and b.op == "const":
        return const(a.val - b.val)
    if _is(b, 0.0):
        return a
    if _is(a, 0.0):
        return neg(b)
    if a is b:
        return ZERO
    if b.op == "neg":
        return add(a, b.a)
    return _mk("sub", a, b)


def mul(a: Node, b: Node) -> Node:
    if a.op == "const" and b.op == "const":
        return const(a.val * b.val)
    if _is(a, 0.0) or _is(b, 0.0):
        return ZERO
    if _is(a, 1.0):
        return b
    if _is(b, 1.0):
        return a
    if _is(a, -1.0):
        return neg(b)
    if _is(b, -1.0):
        return neg(a)
    if a.op == "neg" and b.op == "neg":
        return mul(a.a, b.a)
    if a.op == "neg":
        return neg(mul(a.a, b))
    if b.op == "neg":
        return neg(mul(a, b.a))
    if a is b:
        return unary("sq", a)
    if not _before(a, b):
        a, b = b, a
    return _mk("mul", a, b)


def div(a: Node, b: Node) -> Node:
    if b.op == "const":
        if b.val == 1.0:
            return a
        if a.op == "const":
            return const(a.val / b.val)
        if b.val == -1.0:
            return neg(a)
    if _is(a, 0.0):
        return ZERO
    if a.op == "neg":
        return neg(div(a.a, b))
    return _mk("div", a, b)


def power(a: Node, b: Node) -> Node:
    if b.op == "const":
        e = b.val
        if e == 0.0:
            return ONE
        if e == 1.0:
            return a
        if e == 2.0:
            return unary("sq", a)
        if e == -1.0:
            return div(ONE, a)
        if e == 0.5:
            return unary("sqrt", a)
        if a.op == "const":
            return const(a.val ** e)
        if e == 3.0:
            return mul(unary("sq", a), a)
    return _mk("pow", a, b)


def binary(op: str, a: Node, b: Node) -> Node:
    if op == "add":
        return add(a, b)
    if op == "sub":
        return sub(a, b)
    if op == "mul":
        return mul(a, b)
    if op == "div":
        return div(a, b)
    if op == "pow":
        return power(a, b)
    if a.op == "const" and b.op == "const":
        f = {"fmin": min, "fmax": max, "atan2": math.atan2}[op]
        return const(f(a.val, b.val))
    return _mk(op, a, b)


# --------------------------------------------------------------------------- traversal

def topo(outputs: Iterable[Node]) -> List[Node]:
    """Nodes reachable from `outputs`, children before parents (iterative DFS)."""
    seen = set()
    order: List[Node] = []
    for root in outputs:
        if root.idx in seen:
            continue
        stack = [(root, False)]
        while stack:
            n, done = stack.pop()
            if done:
                order.append(n)
                continue
            if n.idx in seen:
                continue
            seen.add(n.idx)
            stack.append((n, True))
            if n.b is not None and n.b.idx not in seen:
                stack.append((n.b, False))
            if n.a is not None and n.a.idx not in seen:
                stack.append((n.a, False))
    return order


def depends_on(outputs: Iterable[Node], syms: Iterable[Node]) -> bool:
    ids = {s.idx for s in syms}
    return any(n.op == "sym" and n.idx in ids for n in topo(outputs))


def free_symbols(outputs: Iterable[Node]) -> List[Node]:
    return [n for n in topo(outputs) if n.op == "sym"]


# --------------------------------------------------------------------------- AD

def _partials(n: Node) -> Tuple[Optional[Node], Optional[Node]]:
    """d n / d n.a , d n / d n.b as DAG nodes."""
    op, a, b = n.op, n.a, n.b
    if op == "add":
        return ONE, ONE
    if op == "sub":
        return ONE, MONE
    if op == "mul":
        return b, a
    if op == "div":
        return div(ONE, b), neg(div(n, b))
    if op == "neg":
        return MONE, None
    if op == "sq":
        return mul(TWO, a), None
    if op == "sqrt":
        return div(const(0.5), n), None
    if op == "exp":
        return n, None
    if op == "log":
        return div(ONE, a), None
    if op == "sin":
        return unary("cos", a), None
    if op == "cos":
        return neg(unary("sin", a)), None
    if op == "tan":
        return add(ONE, unary("sq", n)), None
    if op == "tanh":
        return sub(ONE, unary("sq", n)), None
    if op == "sinh":
        return unary("cosh", a), None
    if op == "cosh":
        return unary("sinh", a), None
    if op == "asin":
        return div(ONE, unary("sqrt", sub(ONE, unary("sq", a)))), None
    if op == "acos":
        return neg(div(ONE, unary("sqrt", sub(ONE, unary("sq", a))))), None
    if op == "atan":
        return div(ONE, add(ONE, unary("sq", a))), None
    if op == "fabs":
        return unary("sign", a), None
    if op == "sign":
        return ZERO, None
    if op == "inv":
        return neg(unary("sq", n)), None
    if op == "pow":
        if b.op == "const":
            return mul(b, power(a, const(b.val - 1.0))), ZERO
        return mul(b, power(a, sub(b, ONE))), mul(n, unary("log", a))
    if op in ("fmin", "fmax"):
        # CasADi's rule (casadi/core/calculus.hpp, OP_FMIN / OP_FMAX): the partial derivatives are (a <= b) / ((a <= b) + (b <= a)) and its
        # mirror - one for the selected argument, zero for the other, one half each at a tie; here through the existing sign() node
        sg = unary("sign", sub(b, a) if op == "fmin" else sub(a, b))
        half = const(0.5)
        return mul(half, add(ONE, sg)), mul(half, sub(ONE, sg))
    if op == "atan2":
        den = add(unary("sq", a), unary("sq", b))
        return div(b, den), neg(div(a, den))
    raise NotImplementedError(op)


def reverse_gradient(out: Node, wrt: Sequence[Node]) -> List[Node]:
    """d out / d wrt[i] by one reverse sweep (symbolic adjoints)."""
    order = topo([out])
    adj: Dict[int, Node] = {out.idx: ONE}
    for n in reversed(order):
        bar = adj.get(n.idx)
        if bar is None or n.a is None or _is(bar, 0.0):
            continue
        pa, pb = _partials(n)
        ca = mul(bar, pa)
        prev = adj.get(n.a.idx)
        adj[n.a.idx] = ca if prev is None else add(prev, ca)
        if n.b is not None:
            cb = mul(bar, pb)
            prev = adj.get(n.b.idx)
            adj[n.b.idx] = cb if prev is None else add(prev, cb)
    return [adj.get(w.idx, ZERO) for w in wrt]


def forward_jacobian(outs: Sequence[Node], wrt: Sequence[Node]) -> List[List[Node]]:
    """J[i][j] = d outs[i] / d wrt[j]; one forward sweep carrying sparse tangent dicts."""
    order = topo(outs)
    col = {w.idx: j for j, w in enumerate(wrt)}
    tan: Dict[int, Dict[int, Node]] = {}
    for n in order:
        if n.op == "sym":
            j = col.get(n.idx)
            tan[n.idx] = {j: ONE} if j is not None else {}
            continue
        if n.a is None:
            tan[n.idx] = {}
            continue
        ta = tan[n.a.idx]
        tb = tan[n.b.idx] if n.b is not None else {}
        if not ta and not tb:
            tan[n.idx] = {}
            continue
        pa, pb = _partials(n)
        t: Dict[int, Node] = {}
        for j, v in ta.items():
            t[j] = mul(pa, v)
        for j, v in tb.items():
            c = mul(pb, v)
            t[j] = add(t[j], c) if j in t else c
        tan[n.idx] = {j: v for j, v in t.items() if not _is(v, 0.0)}
    nw = len(wrt)
    return [[tan[o.idx].get(j, ZERO) for j in range(nw)] for o in outs]


def substitute_nodes(outs: Sequence[Node], mapping: Dict[int, Node]) -> List[Node]:
    """Rebuild `outs` with symbols (by idx) replaced by nodes."""
    memo: Dict[int, Node] = dict(mapping)
    for n in topo(outs):
        if n.idx in memo:
            continue
        if n.a is None:
            memo[n.idx] = n
        elif n.b is None:
            memo[n.idx] = unary(n.op, memo[n.a.idx])
        else:
            memo[n.idx] = binary(n.op, memo[n.a.idx], memo[n.b.idx])
    return [memo[o.idx] for o in outs]


# --------------------------------------------------------------------------- matrix wrapper

Scalar = Union[int, float, np.floating, np.integer]


def _as_node(v) -> Node:
    if isinstance(v, Node):
        return v
    if isinstance(v, SX):
        if v.shape != (1, 1):
            raise ValueError("expected a scalar expression")
        return v.data[0]
    if isinstance(v, DM):
        return const(float(np.asarray(v.arr).reshape(-1)[0]))
    return const(float(v))


class SX:
    """Dense matrix of scalar DAG nodes, column-major (CasADi convention)."""
    __array_priority__ = 1000
    __array_ufunc__ = None  # let `ndarray @ SX`, `ndarray * SX` defer to the reflected ops

    def __init__(self, data=0.0, shape: Optional[Tuple[int, int]] = None):
        if isinstance(data, SX):
            self.data, self.shape = list(data.data), data.shape
            return
        if isinstance(data, Node):
            self.data, self.shape = [data], (1, 1)
            return
        if isinstance(data, DM):
            data = data.arr
        if shape is not None and isinstance(data, list) and (not data or isinstance(data[0], Node)):
            assert len(data) == shape[0] * shape[1]
            self.data, self.shape = data, tuple(shape)
            return
        if isinstance(data, (list, tuple)) and data and any(isinstance(d, (SX, Node)) for d in data):
            nodes = [_as_node(d) for d in data]
            self.data, self.shape = nodes, (len(nodes), 1)
            return
        arr = np.atleast_1d(np.asarray(data, dtype=float))
        if arr.ndim == 1:
            arr = arr.reshape(-1, 1)
        self.shape = arr.shape
        self.data = [const(v) for v in arr.flatten(order="F")]

    # -- constructors
    @staticmethod
    def sym(name: str, n: Union[int, Tuple[int, int]] = 1, m: int = 1) -> "SX":
        if isinstance(n, (tuple, list)):
            n, m = n
        if n * m == 1:
            return SX([symbol(name)], (n, m))
        return SX([symbol(f"{name}_{i}") for i in range(n * m)], (n, m))

    @staticmethod
    def zeros(n=1, m=1) -> "SX":
        if isinstance(n, (tuple, list)):
            n, m = n
        return SX([ZERO] * (n * m), (n, m))

    @staticmethod
    def ones(n=1, m=1) -> "SX":
        if isinstance(n, (tuple, list)):
            n, m = n
        return SX([ONE] * (n * m), (n, m))

    @staticmethod
    def eye(n) -> "SX":
        return SX([ONE if i == j else ZERO for j in range(n) for i in range(n)], (n, n))

    # -- shape helpers
    def size1(self):
        return self.shape[0]

    def size2(self):
        return self.shape[1]

    def numel(self):
        return self.shape[0] * self.shape[1]

    def size(self, axis=None):
        if axis is None:
            return self.shape
        return self.shape[axis - 1]

    def __len__(self):
        return self.shape[0]

    def is_scalar(self):
        return self.shape == (1, 1)

    def nodes(self) -> List[Node]:
        return list(self.data)

    def is_constant(self):
        return all(d.op == "const" for d in self.data)

    def to_numpy(self) -> np.ndarray:
        if not self.is_constant():
            raise ValueError("expression is not constant")
        return np.array([d.val for d in self.data]).reshape(self.shape, order="F")

    def __float__(self):
        return float(self.to_numpy().reshape(-1)[0])

    @property
    def T(self) -> "SX":
        n, m = self.shape
        return SX([self.data[i + j * n] for i in range(n) for j in range(m)], (m, n))

    def reshape(self, shape) -> "SX":
        assert shape[0] * shape[1] == self.numel()
        return SX(list(self.data), tuple(shape))

    # -- indexing
    def _rc(self, key):
        n, m = self.shape
        if isinstance(key, tuple):
            r, c = key
        else:
            if m == 1 or n == 1 or isinstance(key, (int, np.integer)):
                # linear (column-major) indexing
                idx = np.arange(n * m)[key]
                return np.atleast_1d(idx), None
            r, c = key, slice(None)
        rows = np.atleast_1d(np.arange(n)[r])
        cols = np.atleast_1d(np.arange(m)[c])
        return rows, cols

    def __getitem__(self, key) -> "SX":
        rows, cols = self._rc(key)
        if cols is None:
            if self.shape[0] == 1 and self.shape[1] > 1:
                return SX([self.data[i] for i in rows], (1, len(rows)))
            return SX([self.data[i] for i in rows], (len(rows), 1))
        n = self.shape[0]
        return SX([self.data[i + j * n] for j in cols for i in rows], (len(rows), len(cols)))

    def __setitem__(self, key, value):
        rows, cols = self._rc(key)
        v = value if isinstance(value, SX) else SX(value)
        if cols is None:
            src = v.data if v.numel() == len(rows) else v.data * len(rows)
            for k, i in enumerate(rows):
                self.data[i] = src[k]
            return
        n = self.shape[0]
        cnt = len(rows) * len(cols)
        src = v.data if v.numel() == cnt else v.data * cnt
        k = 0
        for j in cols:
            for i in rows:
                self.data[i + j * n] = src[k]
                k += 1

    def __iter__(self):
        raise TypeError("SX is not iterable; use .nodes() or index explicitly")

    # -- arithmetic
    def _ew(self, other, fn) -> "SX":
        o = other if isinstance(other, SX) else SX(other)
        if self.shape == o.shape:
            return SX([fn(a, b) for a, b in zip(self.data, o.data)], self.shape)
        if o.numel() == 1:
            b = o.data[0]
            return SX([fn(a, b) for a in self.data], self.shape)
        if self.numel() == 1:
            a = self.data[0]
            return SX([fn(a, b) for b in o.data], o.shape)
        raise ValueError(f"shape mismatch {self.shape} vs {o.shape}")

    def __add__(self, o):
        return self._ew(o, add)

    def __radd__(self, o):
        return SX(o)._ew(self, add)

    def __sub__(self, o):
        return self._ew(o, sub)

    def __rsub__(self, o):
        return SX(o)._ew(self, sub)

    def __mul__(self, o):
        return self._ew(o, mul)

    def __rmul__(self, o):
        return SX(o)._ew(self, mul)

    def __truediv__(self, o):
        return self._ew(o, div)

    def __rtruediv__(self, o):
        return SX(o)._ew(self, div)

    def __pow__(self, o):
        return self._ew(o, power)

    def __rpow__(self, o):
        return SX(o)._ew(self, power)

    def __neg__(self):
        return SX([neg(a) for a in self.data], self.shape)

    def __pos__(self):
        return self

    def __matmul__(self, o):
        o = o if isinstance(o, SX) else SX(o)
        return mtimes(self, o)

    def __rmatmul__(self, o):
        return mtimes(SX(o), self)

    def __repr__(self):
        if self.shape == (1, 1):
            return f"SX({self.data[0]!r})"
        return f"SX({self.shape[0]}x{self.shape[1]})"

    def __hash__(self):
        return id(self)


class DM:
    """Numeric dense matrix with the tiny part of casadi.DM the reference touches."""
    __array_priority__ = 900

    def __init__(self, data=0.0, m=None):
        if isinstance(data, DM):
            data = data.arr
        if m is not None and np.isscalar(data):
            self.arr = np.zeros((int(data), int(m)))
            return
        a = np.array(data, dtype=float)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        self.arr = a

    @property
    def shape(self):
        return self.arr.shape

    def full(self):
        return np.array(self.arr)

    def __array__(self, dtype=None, copy=None):
        return self.arr if dtype is None else self.arr.astype(dtype)

    def __float__(self):
        return float(self.arr.reshape(-1)[0])

    def __getitem__(self, k):
        return DM(self.arr.reshape(-1, order="F")[k] if not isinstance(k, tuple) else self.arr[k])

    def __mul__(self, o):
        if isinstance(o, SX):
            return SX(self.arr) * o
        return DM(self.arr * np.asarray(o))

    __rmul__ = __mul__

    def __truediv__(self, o):
        return DM(self.arr / np.asarray(o))

    def __add__(self, o):
        if isinstance(o, SX):
            return SX(self.arr) + o
        return DM(self.arr + np.asarray(o))

    __radd__ = __add__

    def __sub__(self, o):
        if isinstance(o, SX):
            return SX(self.arr) - o
        return DM(self.arr - np.asarray(o))

    def __neg__(self):
        return DM(-self.arr)

    def __repr__(self):
        return f"DM({self.arr.tolist()})"


# --------------------------------------------------------------------------- free functions

def _sx(v) -> SX:
    return v if isinstance(v, SX) else SX(v)


def vertcat(*args) -> SX:
    parts = [_sx(a) for a in args if not (isinstance(a, (list, tuple)) and len(a) == 0)]
    parts = [p for p in parts if p.numel() > 0]
    if not parts:
        return SX([], (0, 1))
    m = parts[0].shape[1]
    assert all(p.shape[1] == m for p in parts), "vertcat: column mismatch"
    n = sum(p.shape[0] for p in parts)
    data = []
    for j in range(m):
        for p in parts:
            pn = p.shape[0]
            data.extend(p.data[j * pn:(j + 1) * pn])
    return SX(data, (n, m))


def vertsplit(x, incr: int = 1) -> list:
    """`casadi.vertsplit(x, incr)`: the row blocks of x (numeric input -> DM blocks, symbolic -> SX blocks)."""
    if isinstance(x, SX):
        return [x[i:i + incr, :] for i in range(0, x.shape[0], incr)]
    a = x.arr if isinstance(x, DM) else np.asarray(x, dtype=float)
    a = a.reshape(-1, 1) if a.ndim < 2 else a
    return [DM(a[i:i + incr, :]) for i in range(0, a.shape[0], incr)]


def horzcat(*args) -> SX:
    parts = [_sx(a) for a in args]
    parts = [p for p in parts if p.numel() > 0]
    if not parts:
        return SX([], (1, 0))
    n = parts[0].shape[0]
    assert all(p.shape[0] == n for p in parts), "horzcat: row mismatch"
    data = []
    for p in parts:
        data.extend(p.data)
    return SX(data, (n, sum(p.shape[1] for p in parts)))


def mtimes(a, b) -> SX:
    a, b = _sx(a), _sx(b)
    if a.numel() == 1 or b.numel() == 1:
        return a * b
    n, k = a.shape
    k2, m = b.shape
    assert k == k2, f"mtimes: {a.shape} x {b.shape}"
    out = []
    for j in range(m):
        for i in range(n):
            acc = ZERO
            for l in range(k):
                acc = add(acc, mul(a.data[i + l * n], b.data[l + j * k]))
            out.append(acc)
    return SX(out, (n, m))


def sum1(a) -> SX:
    a = _sx(a)
    n, m = a.shape
    out = []
    for j in range(m):
        acc = ZERO
        for i in range(n):
            acc = add(acc, a.data[i + j * n])
        out.append(acc)
    return SX(out, (1, m))


def sum2(a) -> SX:
    return sum1(_sx(a).T).T


def sumsqr(a) -> SX:
    a = _sx(a)
    acc = ZERO
    for d in a.data:
        acc = add(acc, unary("sq", d))
    return SX(acc)


def dot(a, b) -> SX:
    a, b = _sx(a), _sx(b)
    acc = ZERO
    for x, y in zip(a.data, b.data):
        acc = add(acc, mul(x, y))
    return SX(acc)


def _unary_fn(op):
    def f(x):
        if isinstance(x, SX):
            return SX([unary(op, d) for d in x.data], x.shape)
        if isinstance(x, DM):
            return DM(getattr(np, {"fabs": "abs", "asin": "arcsin", "acos": "arccos",
                                   "atan": "arctan"}.get(op, op))(x.arr))
        return getattr(np, {"fabs": "abs", "asin": "arcsin", "acos": "arccos",
                            "atan": "arctan"}.get(op, op))(x)
    f.__name__ = op
    return f


exp = _unary_fn("exp")
log = _unary_fn("log")
sqrt = _unary_fn("sqrt")
sin = _unary_fn("sin")
cos = _unary_fn("cos")
tan = _unary_fn("tan")
tanh = _unary_fn("tanh")
sinh = _unary_fn("sinh")
cosh = _unary_fn("cosh")
asin = _unary_fn("asin")
acos = _unary_fn("acos")
atan = _unary_fn("atan")
fabs = _unary_fn("fabs")
sign = _unary_fn("sign")


def fmin(a, b):
    return _sx(a)._ew(b, lambda x, y: binary("fmin", x, y))


def fmax(a, b):
    return _sx(a)._ew(b, lambda x, y: binary("fmax", x, y))


def atan2(a, b):
    return _sx(a)._ew(b, lambda x, y: binary("atan2", x, y))


def jacobian(expr, wrt) -> SX:
    expr, wrt = _sx(expr), _sx(wrt)
    J = forward_jacobian(expr.data, wrt.data)
    n, m = expr.numel(), wrt.numel()
    return SX([J[i][j] for j in range(m) for i in range(n)], (n, m))


def gradient(expr, wrt) -> SX:
    expr, wrt = _sx(expr), _sx(wrt)
    assert expr.numel() == 1
    return SX(reverse_gradient(expr.data[0], wrt.data), (wrt.numel(), 1))


def hessian(expr, wrt) -> Tuple[SX, SX]:
    g = gradient(expr, wrt)
    return jacobian(g, wrt), g


def substitute(expr, old, new) -> SX:
    expr, old, new = _sx(expr), _sx(old), _sx(new)
    assert old.numel() == new.numel()
    mapping = {o.idx: n for o, n in zip(old.data, new.data)}
    return SX(substitute_nodes(expr.data, mapping), expr.shape)


# --------------------------------------------------------------------------- evaluation / emitters

_C_UNARY = {"neg": "-({a})", "sq": "(({a})*({a}))", "sqrt": "sqrt({a})", "exp": "exp({a})",
            "log": "log({a})", "sin": "sin({a})", "cos": "cos({a})", "tan": "tan({a})",
            "tanh": "tanh({a})", "fabs": "fabs({a})", "sinh": "sinh({a})", "cosh": "cosh({a})",
            "asin": "asin({a})", "acos": "acos({a})", "atan": "atan({a})",
            "sign": "((({a})>0.0)-(({a})<0.0))", "inv": "(1.0/({a}))"}
_C_BINARY = {"add": "{a}+{b}", "sub": "{a}-{b}", "mul": "{a}*{b}", "div": "{a}/{b}",
             "pow": "pow({a},{b})", "fmin": "fmin({a},{b})", "fmax": "fmax({a},{b})",
             "atan2": "atan2({a},{b})"}
_NP_UNARY = dict(_C_UNARY, sqrt="np.sqrt({a})", exp="np.exp({a})", log="np.log({a})",
                 sin="np.sin({a})", cos="np.cos({a})", tan="np.tan({a})", tanh="np.tanh({a})",
                 fabs="np.abs({a})", sinh="np.sinh({a})", cosh="np.cosh({a})",
                 asin="np.arcsin({a})", acos="np.arccos({a})", atan="np.arctan({a})",
                 sign="np.sign({a})")
_NP_BINARY = dict(_C_BINARY, pow="np.power({a},{b})", fmin="np.minimum({a},{b})",
                  fmax="np.maximum({a},{b})", atan2="np.arctan2({a},{b})")


def _npfloat(v: float) -> str:
    if math.isinf(v):
        return "np.inf" if v > 0 else "(-np.inf)"
    if math.isnan(v):
        return "np.nan"
    return repr(float(v))


def _cfloat(v: float) -> str:
    if math.isinf(v):
        return "INFINITY" if v > 0 else "(-INFINITY)"
    if math.isnan(v):
        return "NAN"
    s = repr(float(v))
    if "e" not in s and "." not in s and "n" not in s:
        s += ".0"
    return s


def emit_c(outputs: Sequence[Tuple[str, Node]], inputs: Dict[int, str],
           indent: str = "  ", skip_zero: bool = False, accumulate: bool = False, numpy: bool = False) -> str:
    """Straight-line C for `outputs` = [(lvalue, node)].

    `inputs` maps symbol idx -> C rvalue (e.g. "x[3]").  Every interior node that is
    used more than once, or is a transcendental, gets a named temporary; single-use
    arithmetic is inlined so the compiler sees FMA-able trees.
    """
    unary_tab, binary_tab = (_NP_UNARY, _NP_BINARY) if numpy else (_C_UNARY, _C_BINARY)
    decl, end = ("", "") if numpy else ("const double ", ";")
    nodes = topo([n for _, n in outputs])
    uses: Dict[int, int] = {}
    for n in nodes:
        for c in (n.a, n.b):
            if c is not None:
                uses[c.idx] = uses.get(c.idx, 0) + 1
    for _, n in outputs:
        uses[n.idx] = uses.get(n.idx, 0) + 1
    name: Dict[int, str] = {}
    lines: List[str] = []
    tcount = 0

    def ref(n: Node) -> str:
        return name[n.idx]

    for n in nodes:
        if n.op == "const":
            lit = _npfloat(n.val) if numpy else _cfloat(n.val)
            name[n.idx] = lit if n.val >= 0 else f"({lit})"
            continue
        if n.op == "sym":
            if n.idx not in inputs:
                raise KeyError(f"free symbol {n.val} has no binding")
            name[n.idx] = inputs[n.idx]
            continue
        if n.b is None:
            ex = unary_tab[n.op].format(a=ref(n.a))
        else:
            ex = binary_tab[n.op].format(a=ref(n.a), b=ref(n.b))
        inline = uses.get(n.idx, 0) <= 1 and n.op in ("add", "sub", "mul", "neg", "div") and len(ex) < 200
        if inline:
            name[n.idx] = f"({ex})"
        else:
            tn = f"t{tcount}"
            tcount += 1
            lines.append(f"{indent}{decl}{tn} = {ex}{end}")
            name[n.idx] = tn
    for lv, n in outputs:
        if skip_zero and _is(n, 0.0):
            continue
        lines.append(f"{indent}{lv} {'+=' if accumulate else '='} {ref(n)}{end}")
    return "\n".join(lines)


class Function:
    """Numeric/symbolic callable over named SX inputs (subset of casadi.Function)."""

    def __init__(self, name: str, inputs: Sequence[SX], outputs: Sequence[SX],
                 in_names: Optional[Sequence[str]] = None, out_names: Optional[Sequence[str]] = None):
        self.name = name
        self.inputs = [_sx(i) for i in inputs]
        self.outputs = [_sx(o) for o in outputs]
        self.in_names = list(in_names) if in_names else [f"i{k}" for k in range(len(self.inputs))]
        self.out_names = list(out_names) if out_names else [f"o{k}" for k in range(len(self.outputs))]
        for i in self.inputs:
            for d in i.data:
                if d.op != "sym":
                    raise ValueError("Function inputs must be purely symbolic")
        self._np = None

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_np"] = None                     # the compiled numpy callable is rebuilt on first use after unpickling
        return st

    def n_in(self):
        return len(self.inputs)

    def n_out(self):
        return len(self.outputs)

    def free_symbols(self) -> List[Node]:
        bound = {d.idx for i in self.inputs for d in i.data}
        out_nodes = [d for o in self.outputs for d in o.data]
        return [s for s in free_symbols(out_nodes) if s.idx not in bound]

    def _compile_numpy(self):
        binds = {}
        for k, i in enumerate(self.inputs):
            for e, d in enumerate(i.data):
                binds[d.idx] = f"a{k}[{e}]"
        outs = []
        for k, o in enumerate(self.outputs):
            for e, d in enumerate(o.data):
                outs.append((f"r{k}[{e}]", d))
        # numpy source straight from the operator tables (_NP_UNARY / _NP_BINARY): every operator of the DAG has an
        # array-valued counterpart there (arcsin/arccos/arctan, np.sign, ...), nothing is translated from C text
        body = emit_c(outs, binds, indent="    ", numpy=True)
        src = "def _f(" + ",".join(f"a{k}" for k in range(len(self.inputs))) + "," + \
              ",".join(f"r{k}" for k in range(len(self.outputs))) + "):\n"
        src += body if body.strip() else "    pass"
        src += "\n"
        ns = {"np": np}
        exec(compile(src, f"<sym:{self.name}>", "exec"), ns)
        self._np = ns["_f"]

    def eval(self, *args) -> List[np.ndarray]:
        """Numeric evaluation.  Each arg may carry trailing batch dims: shape (numel, ...)."""
        if self._np is None:
            self._compile_numpy()
        ins = []
        batch = ()
        for a, i in zip(args, self.inputs):
            arr = np.asarray(a.arr if isinstance(a, DM) else a, dtype=float)
            if arr.size == i.numel():
                arr = arr.reshape(-1, order="F")
            elif arr.shape[0] != i.numel():
                raise ValueError(f"{self.name}: input of shape {arr.shape} for {i.shape}")
            if arr.ndim > 1:
                batch = arr.shape[1:]
            ins.append(arr)
        outs = [np.zeros((o.numel(),) + batch) for o in self.outputs]
        self._np(*ins, *outs)
        return outs

    def __call__(self, *args):
        if len(args) != len(self.inputs):
            raise TypeError(f"{self.name}: expected {len(self.inputs)} inputs, got {len(args)}")
        if any(isinstance(a, SX) and not a.is_constant() for a in args):
            mapping = {}
            for a, i in zip(args, self.inputs):
                a = _sx(a)
                if a.numel() != i.numel():
                    raise ValueError(f"{self.name}: input size {a.shape} != {i.shape}")
                for d, s in zip(a.data, i.data):
                    mapping[s.idx] = d
            res = []
            for o in self.outputs:
                res.append(SX(substitute_nodes(o.data, mapping), o.shape))
            return res[0] if len(res) == 1 else res
        num = [a.to_numpy().reshape(-1, order="F") if isinstance(a, SX) else
               np.asarray(a.arr if isinstance(a, DM) else a, dtype=float).reshape(-1, order="F")
               for a in args]
        outs = self.eval(*num)
        res = [DM(o.reshape(s.shape, order="F")) for o, s in zip(outs, self.outputs)]
        return res[0] if len(res) == 1 else res
