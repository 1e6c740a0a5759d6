#!/usr/bin/env python3
"""Container-only generator tooling: execute the reference's OWN C++ statements without a CUDA toolchain.

The reference (plancherb1/parallel-DDP) is CUDA C++ that cannot be compiled in this image (it includes cuda_runtime.h, cublas_v2.h, cusolverDn.h; writing
stand-ins for them is not allowed).  Its hot path, however, is written in a small, regular subset of C: counted loops, index arithmetic on flat arrays,
templated helper calls, `#ifdef __CUDA_ARCH__` host / device branches, `__shared__` arrays and `__syncthreads()`.  This module

  1. runs a C preprocessor over the reference sources where they lie (object- and function-like macros, #if / #ifdef / #elif / #else, quoted #include),
     with `__CUDA_ARCH__` defined (device branches) or not (host branches);
  2. parses the function definitions it is asked for and rewrites every statement MECHANICALLY into Python -- one Python statement per C statement, same
     operand order, same index expressions, C integer division / remainder kept, pointers as (buffer, offset) views;
  3. executes them in float64: host functions directly; device functions and __global__ kernels under a SIMT emulation -- one Python generator per CUDA
     thread, every `__syncthreads()` a yield, all threads of a block advanced barrier to barrier, `__shared__` arrays shared by the block, threadIdx /
     blockIdx / blockDim / gridDim as launched.

So the numbers the fixture generators (make_phase_fixtures.py) store are the reference's own formulas evaluated in double precision with the reference's
launch geometry.  Nothing of the reference's text is stored anywhere: this file holds a C-subset translator, the generators hold function NAMES and
stored inputs, the fixtures hold numbers.  /root/reference is read at generation time only.
"""
import math
import os
import re

# ------------------------------------------------------------------------------------------------------------------ tokens
TOKEN_RE = re.compile(r"""
    (?P<num>(?:0[xX][0-9a-fA-F]+|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?)[fFuUlL]*)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<str>"(?:\\.|[^"\\])*")
  | (?P<chr>'(?:\\.|[^'\\])')
  | (?P<op><<<|>>>|<<=|>>=|\+\+|--|->|&&|\|\||==|!=|<=|>=|\+=|-=|\*=|/=|%=|&=|\|=|\^=|<<|>>|::|[-+*/%<>=!&|^~?:;,.(){}\[\]\#])
""", re.X)


def tokenize(text):
    out, pos = [], 0
    for m in TOKEN_RE.finditer(text):
        gap = text[pos:m.start()]
        if gap.strip():
            raise SyntaxError("cannot tokenize: %r" % gap[:40])
        out.append((m.lastgroup, m.group()))
        pos = m.end()
    return out


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group().count("\n"), text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return text.replace("\\\r\n", " ").replace("\\\n", " ")


# ------------------------------------------------------------------------------------------------------------------ preprocessor
class Preprocessor:
    """Enough of cpp for the reference's headers.  `predefined` are -D style definitions that later #defines of the same name do not override
    (the effect of an #ifndef guard around them: the survey's "patched temp copy" of config.cuh, without touching the file)."""

    def __init__(self, root, predefined=None, cuda_arch=False, skip_includes=()):
        self.root, self.macros, self.locked = root, {}, set()
        for k, v in (predefined or {}).items():
            self.macros[k] = (None, tokenize(str(v)))
            self.locked.add(k)
        if cuda_arch:
            self.macros["__CUDA_ARCH__"] = (None, tokenize("600"))
        self.skip = set(skip_includes)
        self.out = []

    # -- macro expansion over a token list
    def expand(self, toks, hide=frozenset()):
        out, i = [], 0
        while i < len(toks):
            kind, val = toks[i]
            if kind == "id" and val in self.macros and val not in hide:
                params, body = self.macros[val]
                if params is None:
                    out += self.expand(body, hide | {val})
                    i += 1
                    continue
                if i + 1 < len(toks) and toks[i + 1][1] == "(":
                    args, j = self._args(toks, i + 1)
                    if len(params) == 1 and len(args) == 0:
                        args = [[]]
                    assert len(args) == len(params), (val, params, args)
                    amap = {p: self.expand(a, hide) for p, a in zip(params, args)}
                    sub = []
                    for k2, v2 in body:
                        sub += amap[v2] if (k2 == "id" and v2 in amap) else [(k2, v2)]
                    out += self.expand(sub, hide | {val})
                    i = j
                    continue
            out.append((kind, val))
            i += 1
        return out

    @staticmethod
    def _args(toks, i):
        assert toks[i][1] == "("
        depth, args, cur, j = 0, [], [], i
        while True:
            v = toks[j][1]
            if v in "([{":
                depth += 1
                if depth > 1:
                    cur.append(toks[j])
            elif v in ")]}":
                depth -= 1
                if depth == 0:
                    if cur or args:
                        args.append(cur)
                    return args, j + 1
                cur.append(toks[j])
            elif v == "," and depth == 1:
                args.append(cur); cur = []
            else:
                cur.append(toks[j])
            j += 1

    def _eval_if(self, expr):
        toks = tokenize(expr)
        res, i = [], 0
        while i < len(toks):                                       # defined(X) / defined X before expansion
            if toks[i][1] == "defined":
                if toks[i + 1][1] == "(":
                    name, i = toks[i + 2][1], i + 4
                else:
                    name, i = toks[i + 1][1], i + 2
                res.append(("num", "1" if name in self.macros else "0"))
            else:
                res.append(toks[i]); i += 1
        toks = self.expand(res)
        py = []
        for kind, v in toks:
            if kind == "id":
                py.append("0")                                     # an identifier that is not a macro evaluates to 0
            elif v == "&&":
                py.append(" and ")
            elif v == "||":
                py.append(" or ")
            elif v == "!":
                py.append(" not ")
            elif kind == "num":
                py.append(re.sub(r"[fFuUlL]+$", "", v))
            else:
                py.append(v)
        return bool(eval("".join(py)))

    def process_file(self, rel):
        path = os.path.join(self.root, rel)
        text = strip_comments(open(path, errors="replace").read())
        stack = []                                                 # (active_before, taken_any, active_now)
        active = True
        for line in text.split("\n"):
            s = line.strip()
            if s.startswith("#"):
                m = re.match(r"#\s*(\w+)\s*(.*)", s)
                d, rest = m.group(1), m.group(2).strip()
                if d in ("if", "ifdef", "ifndef"):
                    if not active:
                        stack.append((False, True, False)); continue
                    c = self._eval_if(rest) if d == "if" else ((rest.split()[0] in self.macros) == (d == "ifdef"))
                    stack.append((True, c, c)); active = c
                elif d == "elif":
                    before, taken, _ = stack[-1]
                    c = before and not taken and self._eval_if(rest)
                    stack[-1] = (before, taken or c, c); active = c
                elif d == "else":
                    before, taken, _ = stack[-1]
                    c = before and not taken
                    stack[-1] = (before, True, c); active = c
                elif d == "endif":
                    before, _, _ = stack.pop(); active = before
                elif not active:
                    continue
                elif d == "define":
                    m2 = re.match(r"(\w+)(\(([^)]*)\))?\s*(.*)", rest)
                    name = m2.group(1)
                    if name in self.locked:
                        continue
                    fn_like = m2.group(2) is not None and rest[len(name):len(name) + 1] == "("
                    if fn_like:
                        params = [p.strip() for p in m2.group(3).split(",") if p.strip()]
                        self.macros[name] = (params, tokenize(m2.group(4)))
                    else:
                        self.macros[name] = (None, tokenize(rest[len(name):]))
                elif d == "undef":
                    self.macros.pop(rest.split()[0], None)
                elif d == "include":
                    m2 = re.match(r'"([^"]+)"', rest)
                    if m2 and os.path.basename(m2.group(1)) not in self.skip:
                        cand = [os.path.join(os.path.dirname(rel), m2.group(1)), m2.group(1)]
                        inc = next((c for c in cand if os.path.exists(os.path.join(self.root, c))), None)
                        if inc:
                            self.process_file(os.path.normpath(inc))
                elif d == "error":
                    raise RuntimeError("#error reached in %s: %s" % (rel, rest))
                continue                                           # pragma and anything else: dropped
            if active and s:
                self.out += self.expand(tokenize(line))
        return self

    def value(self, name):
        """numeric value of an object-like macro (after expansion)"""
        toks = self.expand([("id", name)])
        py = "".join(re.sub(r"[fFuUlL]+$", "", v) if k == "num" else v for k, v in toks)
        return eval(py, {"max": max, "min": min})


# ------------------------------------------------------------------------------------------------------------------ run-time support
class Ptr:
    """A C pointer into a flat buffer: (buffer, offset).  Buffers are Python lists (or any indexable)."""
    __slots__ = ("buf", "off")

    def __init__(self, buf, off=0):
        self.buf, self.off = buf, off

    @staticmethod
    def alloc(n, fill=None):
        return Ptr([REAL(0.0) if fill is None else fill] * int(n), 0)

    def __add__(self, i):
        return Ptr(self.buf, self.off + int(i))
    __radd__ = __add__

    def __sub__(self, i):
        if isinstance(i, Ptr):
            assert i.buf is self.buf
            return self.off - i.off
        return Ptr(self.buf, self.off - int(i))

    def __getitem__(self, i):
        j = self.off + int(i)
        if j < 0:
            raise IndexError("negative offset %d" % j)
        return self.buf[j]

    def __setitem__(self, i, v):
        j = self.off + int(i)
        if j < 0:
            raise IndexError("negative offset %d" % j)
        if type(v) is not F32 and type(self.buf[j]) is F32:      # a store into a float array converts (float32 mode)
            v = F32(v)
        self.buf[j] = v

    def __bool__(self):
        return True


# ------------------------------------------------------------------------------------------------------------------ float32 mode (round 4)
# REAL = float (the default): every T is a Python float, i.e. the reference's statements in double precision -- what phase_fixtures.npz was generated with.
# set_real("f32"): T = F32, a float32 VALUE TYPE with C's usual arithmetic conversions, so that the translated statements evaluate like the reference compiled with
# algType = float, strict IEEE, no contraction:  F32 op F32 -> F32 (one float32 rounding per operation);  F32 op int -> F32 (the int converts to float);  F32 op double
# (a Python float: an unsuffixed literal, a (double) cast, a double variable) -> double;  the result is rounded to float32 where C converts it: on assignment to a T
# variable or array element, as a T argument or return value, in a (T) cast.  sin / cos / atan2 / pow of a float32 are libm's float functions (the overloads C++ picks).
import ctypes
import numpy as _np

_f32 = _np.float32
_np.seterr(all="ignore")


def _mk(v):
    r = F32.__new__(F32)
    r.v = v
    return r


class F32:
    __slots__ = ("v",)

    def __init__(self, x=0.0):
        self.v = x.v if type(x) is F32 else _f32(x)

    def __add__(self, o):
        t = type(o)
        if t is F32: return _mk(self.v + o.v)
        if t is float: return float(self.v) + o
        return _mk(self.v + _f32(o))

    def __radd__(self, o):
        return (o + float(self.v)) if type(o) is float else _mk(_f32(o) + self.v)

    def __sub__(self, o):
        t = type(o)
        if t is F32: return _mk(self.v - o.v)
        if t is float: return float(self.v) - o
        return _mk(self.v - _f32(o))

    def __rsub__(self, o):
        return (o - float(self.v)) if type(o) is float else _mk(_f32(o) - self.v)

    def __mul__(self, o):
        t = type(o)
        if t is F32: return _mk(self.v * o.v)
        if t is float: return float(self.v) * o
        return _mk(self.v * _f32(o))

    def __rmul__(self, o):
        return (o * float(self.v)) if type(o) is float else _mk(_f32(o) * self.v)

    def __truediv__(self, o):
        t = type(o)
        if t is F32: return _mk(self.v / o.v)
        if t is float: return float(_np.float64(self.v) / _np.float64(o))
        return _mk(self.v / _f32(o))

    def __rtruediv__(self, o):
        return float(_np.float64(o) / _np.float64(self.v)) if type(o) is float else _mk(_f32(o) / self.v)

    def __neg__(self): return _mk(-self.v)
    def __pos__(self): return self
    def __abs__(self): return _mk(abs(self.v))
    def __float__(self): return float(self.v)
    def __int__(self): return int(self.v)
    def __bool__(self): return bool(self.v != 0)
    def __repr__(self): return "F32(%r)" % float(self.v)
    def _o(self, o): return float(o.v) if type(o) is F32 else o
    def __eq__(self, o): return float(self.v) == self._o(o)
    def __ne__(self, o): return float(self.v) != self._o(o)
    def __lt__(self, o): return float(self.v) < self._o(o)
    def __le__(self, o): return float(self.v) <= self._o(o)
    def __gt__(self, o): return float(self.v) > self._o(o)
    def __ge__(self, o): return float(self.v) >= self._o(o)
    __hash__ = None


REAL = float


def set_real(kind):
    """"f64" (default) or "f32": the element type T of everything translated and executed from here on"""
    global REAL
    REAL = F32 if kind == "f32" else float
    return REAL


def typed():
    return REAL is F32


_libm = ctypes.CDLL("libm.so.6")
for _n in ("sinf", "cosf", "tanf", "expf", "logf"):
    getattr(_libm, _n).restype = ctypes.c_float; getattr(_libm, _n).argtypes = [ctypes.c_float]
for _n in ("atan2f", "powf"):
    getattr(_libm, _n).restype = ctypes.c_float; getattr(_libm, _n).argtypes = [ctypes.c_float, ctypes.c_float]


def _m1(fname, dbl):
    cf = getattr(_libm, fname)
    def f(x):
        return _mk(_f32(cf(float(x.v)))) if type(x) is F32 else dbl(x)
    return f


t_sin, t_cos, t_tan, t_exp, t_log = _m1("sinf", math.sin), _m1("cosf", math.cos), _m1("tanf", math.tan), _m1("expf", math.exp), _m1("logf", math.log)


def t_sqrt(x):
    return _mk(_np.sqrt(x.v)) if type(x) is F32 else math.sqrt(x)


def t_atan2(y, x):
    if type(y) is F32 and type(x) is F32:
        return _mk(_f32(_libm.atan2f(float(y.v), float(x.v))))
    return math.atan2(float(y), float(x))


def c_div(a, b):
    if isinstance(a, int) and isinstance(b, int) and not isinstance(a, bool):
        q = abs(a) // abs(b)
        return q if (a >= 0) == (b >= 0) else -q
    return a / b


def c_mod(a, b):
    if isinstance(a, int) and isinstance(b, int):
        return a - b * c_div(a, b)
    return math.fmod(a, b)


def c_pow(a, b):
    if type(a) is F32 and type(b) is F32:                        # pow(float, float) -> powf; pow(float, int) and pow(float, double) are double (C++11 promotion)
        return _mk(_f32(_libm.powf(float(a.v), float(b.v))))
    return math.pow(float(a), float(b))


SIZEOF = 8        # sizeof(T), in the unit memset / memcpy byte counts are given in: element counts = bytes / SIZEOF (every buffer here holds one Python number per element)


def c_memset(p, v, nbytes):
    for i in range(int(nbytes) // SIZEOF):
        p[i] = type(p[i])(v)


def c_memcpy(dst, src, nbytes):
    for i in range(int(nbytes) // SIZEOF):
        dst[i] = src[i]


def cu_memcpy(dst, src, nbytes, *rest):
    c_memcpy(dst, src, nbytes)
    return 0


def cu_memset(p, v, nbytes, *rest):
    c_memset(p, v, nbytes)
    return 0


def cu_ok(*a):
    return 0


class Dim3:
    def __init__(self, x=1, y=1, z=1):
        self.x, self.y, self.z = int(x), int(y), int(z)


class Thread:
    __slots__ = ("tix", "tiy", "tiz", "bdx", "bdy", "bdz", "bix", "biy", "biz", "gdx", "gdy", "gdz", "blk")

    def __init__(self, tix=0, tiy=0, bdx=1, bdy=1, bix=0, biy=0, gdx=1, gdy=1, blk=None):
        self.tix, self.tiy, self.tiz, self.bdx, self.bdy, self.bdz = tix, tiy, 0, bdx, bdy, 1
        self.bix, self.biy, self.biz, self.gdx, self.gdy, self.gdz = bix, biy, 0, gdx, gdy, 1
        self.blk = blk


class Block:
    def __init__(self, extern_elems=0):
        self.mem = {}
        self.extern = Ptr.alloc(extern_elems) if extern_elems else None

    def shared(self, key, n, fill=None):
        if key not in self.mem:
            self.mem[key] = Ptr.alloc(n, fill)
        return self.mem[key]


HOST = Thread()


class Struct:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def padd(p, i):
    """&p[i]: C forms the address without touching memory, also for a null p (an optional output the callee then never dereferences)"""
    return None if p is None else p + i


UNSET = object()


def dflt(thunk):
    """a default argument; one that names a macro the configuration does not define (the end-effector weights outside the arm's cost file) is never used there"""
    try:
        return thunk()
    except NameError:
        return None


def tpl(given, names, defaults):
    """template arguments: explicit ones first, then the defaults; T (typename) is always float here"""
    vals = list(given)
    out = []
    for i, (nm, d) in enumerate(zip(names, defaults)):
        out.append(vals[i] if i < len(vals) else d)
    return out


# ------------------------------------------------------------------------------------------------------------------ translator
TYPE_WORDS = {"T", "int", "int64_t", "unsigned", "bool", "float", "double", "auto", "const", "char", "long", "size_t", "threadDesc_t", "dim3", "void", "half", "algType", "struct", "timeval"}
QUALIFIERS = {"__host__", "__device__", "__global__", "__forceinline__", "inline", "static", "extern", "__noinline__", "constexpr"}
SYNC_NAMES = {"__syncthreads"}
MATH = {"sin": "t_sin", "cos": "t_cos", "sqrt": "t_sqrt", "abs": "abs", "fabs": "abs", "pow": "c_pow", "atan2": "t_atan2", "max": "max", "min": "min",
        "exp": "t_exp", "log": "t_log", "tan": "t_tan", "floor": "math.floor", "ceil": "math.ceil", "memset": "c_memset", "memcpy": "c_memcpy"}      # t_*: libm's double functions for a Python float, its float functions for an F32
# host side of the CUDA runtime as far as the reference's host drivers use it: copies and fills act on the emulated buffers (byte counts are multiples of sizeof(T)),
# synchronisation and error queries are no-ops (the emulation is sequential)
CUDA_RT = {"cudaMemcpy": "cu_memcpy", "cudaMemcpyAsync": "cu_memcpy", "cudaMemset": "cu_memset", "cudaMemsetAsync": "cu_memset", "cudaStreamSynchronize": "cu_ok",
           "cudaDeviceSynchronize": "cu_ok", "cudaPeekAtLastError": "cu_ok", "gpuAssert": "cu_ok", "gettimeofday": "cu_ok"}
CUDA_VARS = {("threadIdx", "x"): "_t.tix", ("threadIdx", "y"): "_t.tiy", ("threadIdx", "z"): "_t.tiz", ("blockIdx", "x"): "_t.bix", ("blockIdx", "y"): "_t.biy",
             ("blockIdx", "z"): "_t.biz", ("blockDim", "x"): "_t.bdx", ("blockDim", "y"): "_t.bdy", ("blockDim", "z"): "_t.bdz", ("gridDim", "x"): "_t.gdx",
             ("gridDim", "y"): "_t.gdy", ("gridDim", "z"): "_t.gdz"}


class Param(tuple):
    """(name, stars, default) like before + .words: the parameter's type words (float32 mode converts by-value floating arguments at function entry)"""
    def __new__(cls, t, words):
        o = tuple.__new__(cls, t)
        o.words = words
        return o


class FuncDef:
    def __init__(self, name, tparams, params, body, ret, is_kernel):
        self.name, self.tparams, self.params, self.body, self.ret, self.is_kernel = name, tparams, params, body, ret, is_kernel
        self.pyname = None
        self.syncs = False


def split_top(toks, sep=","):
    parts, cur, depth = [], [], 0
    for t in toks:
        v = t[1]
        if v in "([{":
            depth += 1
        elif v in ")]}":
            depth -= 1
        if v == sep and depth == 0:
            parts.append(cur); cur = []
        else:
            cur.append(t)
    if cur or parts:
        parts.append(cur)
    return parts


def split_template_args(toks):
    """comma split that also respects < > nesting (template parameter lists only)"""
    parts, cur, depth = [], [], 0
    for t in toks:
        v = t[1]
        if v in "([{<":
            depth += 1
        elif v in ")]}>":
            depth -= 1
        if v == "," and depth == 0:
            parts.append(cur); cur = []
        else:
            cur.append(t)
    if cur or parts:
        parts.append(cur)
    return parts


def find_functions(toks):
    """top-level function definitions of a preprocessed token stream -> {name: [FuncDef, ...]}"""
    funcs, i, n = {}, 0, len(toks)
    header = []
    while i < n:
        v = toks[i][1]
        if v == ";":
            header = []; i += 1; continue
        if v == "{":
            depth, j = 1, i + 1
            while depth:
                depth += toks[j][1] == "{"
                depth -= toks[j][1] == "}"
                j += 1
            body = toks[i + 1:j - 1]
            fd = parse_header(header, body)
            if fd:
                funcs.setdefault(fd.name, []).append(fd)
            header = []; i = j; continue
        header.append(toks[i]); i += 1
    return funcs


def parse_header(h, body):
    if not h or not any(t[1] == "(" for t in h):
        return None
    tparams = []
    k = 0
    if h[0][1] == "template":
        assert h[1][1] == "<"
        depth, j = 1, 2
        while depth:
            depth += h[j][1] == "<"
            depth -= h[j][1] == ">"
            j += 1
        for part in split_template_args(h[2:j - 1]):
            if not part:
                continue
            eq = next((x for x, t in enumerate(part) if t[1] == "="), None)
            default = part[eq + 1:] if eq is not None else None
            decl = part[:eq] if eq is not None else part
            tparams.append((decl[-1][1], decl[0][1] in ("typename", "class"), default))
        k = j
    rest = h[k:]
    try:
        po = next(x for x, t in enumerate(rest) if t[1] == "(")
    except StopIteration:
        return None
    if po == 0 or rest[po - 1][0] != "id":
        return None
    name = rest[po - 1][1]
    pre = [t[1] for t in rest[:po - 1]]
    if any(p in ("struct", "class", "typedef", "namespace", "=", "enum") for p in pre) or name in ("if", "for", "while", "switch"):
        return None
    depth, j = 1, po + 1
    while depth:
        depth += rest[j][1] == "("
        depth -= rest[j][1] == ")"
        j += 1
    params = []
    for part in split_top(rest[po + 1:j - 1]):
        if not part or (len(part) == 1 and part[0][1] == "void"):
            continue
        eq = next((x for x, t in enumerate(part) if t[1] == "="), None)
        default = part[eq + 1:] if eq is not None else None
        decl = part[:eq] if eq is not None else part
        ids = [t for t in decl if t[0] == "id"]
        pname = ids[-1][1]
        stars = sum(1 for t in decl if t[1] == "*")
        params.append(Param((pname, stars, default), [t[1] for t in ids[:-1]]))
    ret = [p for p in pre if p not in QUALIFIERS]
    return FuncDef(name, tparams, params, body, ret, "__global__" in pre)


class Translator:
    """One translation unit: preprocessed tokens -> Python functions (lazily, by name, following the call graph)."""

    def __init__(self, toks, device):
        self.device = device
        self.funcs = {k: v for k, v in find_functions(toks).items() if k not in MATH}   # (cudaUtils.h overloads sin / cos / min ... for `half`: libm's are used)
        self.extra_ns = {}                  # run-time hooks of the harness (launch_kernel)
        self.emitted = {}                   # pyname -> source
        self.order = []
        self.ns = None
        for name, lst in self.funcs.items():
            for idx, fd in enumerate(lst):
                fd.pyname = name if len(lst) == 1 else "%s__%d" % (name, len(fd.params))
        self._compute_syncs()

    # which functions reach a barrier (device mode only): fixed point over the call graph
    def _compute_syncs(self):
        if not self.device:
            return
        calls = {}
        for name, lst in self.funcs.items():
            for fd in lst:
                ids = {t[1] for t in fd.body if t[0] == "id"}
                calls[fd] = ids
                fd.syncs = bool(ids & SYNC_NAMES)
        changed = True
        while changed:
            changed = False
            for fd, ids in calls.items():
                if fd.syncs:
                    continue
                for nm in ids:
                    if nm in self.funcs and any(g.syncs for g in self.funcs[nm]):
                        fd.syncs = True; changed = True
                        break

    def overload(self, name, nargs):
        lst = self.funcs[name]
        if len(lst) == 1:
            return lst[0]
        ok = [fd for fd in lst if sum(1 for p in fd.params if p[2] is None) <= nargs <= len(fd.params)]
        if not ok:                                                 # the reference calls some plug-ins with more arguments than their (older) definitions
            ok = sorted(lst, key=lambda fd: -len(fd.params))[:1]   # take: the widest definition absorbs them (extra arguments are ignored)
        return sorted(ok, key=lambda fd: len(fd.params))[0]

    def require(self, fd):
        if fd.pyname in self.emitted:
            return
        self.emitted[fd.pyname] = None                             # guards recursion
        src = FuncTranslator(self, fd).source()
        self.emitted[fd.pyname] = src
        self.order.append(fd.pyname)

    def build(self, names):
        for nm in names:
            for fd in self.funcs[nm]:
                self.require(fd)
        ns = {"t_sin": t_sin, "t_cos": t_cos, "t_tan": t_tan, "t_exp": t_exp, "t_log": t_log, "t_sqrt": t_sqrt, "t_atan2": t_atan2, "F32": F32, "REAL": REAL, "math": math, "Ptr": Ptr, "c_div": c_div, "c_mod": c_mod, "c_pow": c_pow, "tpl": tpl, "Struct": Struct, "HOST": HOST, "UNSET": UNSET, "dflt": dflt, "c_memset": c_memset, "c_memcpy": c_memcpy, "padd": padd, "cu_memcpy": cu_memcpy, "cu_memset": cu_memset, "cu_ok": cu_ok,
              "Dim3": Dim3, "cudaMemcpyHostToDevice": 1, "cudaMemcpyDeviceToHost": 2, "cudaMemcpyDeviceToDevice": 3, "__FILE__": 0, "__LINE__": 0}
        ns.update(self.extra_ns)
        for py in self.order:
            exec(self.emitted[py], ns)
        self.ns = ns
        return ns

    def source(self):
        return "\n\n".join(self.emitted[p] for p in self.order)


class FuncTranslator:
    def __init__(self, tu, fd):
        self.tu, self.fd = tu, fd
        self.lines = []
        self.tmp = 0
        self.scalars = {p[0] for p in fd.params if p[1] == 0}   # scalar parameters and declared scalar locals (for &name out-arguments)
        self.loop_incr = []               # stack of increment statements of the enclosing for-loops (for `continue`)
        self.tnames = {tp[0] for tp in fd.tparams if tp[1]}        # typename parameters only (T): integer template parameters are values
        self.vtypes = {}                  # float32 mode: scalar name -> the conversion its C type applies on assignment ("T", "F32", "float")

    def caster(self, words):
        """the Python conversion of a C floating type: a typename parameter -> that (run-time) type, algType / T outside a template -> REAL, float -> F32, double -> float.
        Outside float32 mode everything is a Python float, as it always was."""
        if not typed():
            return "float"
        for w in words:
            if w in self.tnames:
                return w
        if any(w in ("T", "algType") for w in words):
            return "REAL"
        if "float" in words:
            return "F32"
        return "float"

    # ---------------------------------------------------------------- emit helpers
    def emit(self, ind, s):
        self.lines.append("    " * ind + s)

    def newtmp(self):
        self.tmp += 1
        return "_v%d" % self.tmp

    def source(self):
        fd = self.fd
        ps, late = [], []
        for pname, stars, default in fd.params:
            if default is not None:
                ps.append("%s=UNSET" % pname)                       # C++ default arguments are evaluated at the call: so are these
                late.append("if %s is UNSET: %s = dflt(lambda: %s)" % (pname, pname, self.expr(default, 1, pre_ok=False)))
            else:
                ps.append(pname)
        head = "def %s(_t, _tp%s, *_extra):" % (fd.pyname, "".join(", " + p for p in ps))
        self.lines = [head]
        for l in late:
            self.emit(1, l)
        if fd.tparams:
            names = [tp[0] for tp in fd.tparams]
            defaults = ["REAL" if tp[1] else (self.expr(tp[2], 1, pre_ok=False) if tp[2] is not None else "None") for tp in fd.tparams]
            self.emit(1, "%s = tpl(_tp, %r, [%s])" % (", ".join(names) + ("," if len(names) == 1 else ""), names, ", ".join(defaults)))
        if typed():                                                 # by-value floating parameters convert to their type (T rho, T alpha, T Q1 = _Q1 ...)
            for prm in fd.params:
                words = getattr(prm, "words", [])
                if prm[1] == 0 and any(w in self.tnames or w in ("T", "algType", "float", "double") for w in words):
                    c = self.caster(words)
                    self.vtypes[prm[0]] = c
                    self.emit(1, "if %s is not None: %s = %s(%s)" % (prm[0], prm[0], c, prm[0]))
        n0 = len(self.lines)
        self.block(fd.body, 1)
        if len(self.lines) == n0:
            self.emit(1, "pass")
        if self.tu.device and fd.syncs and not any("yield" in l for l in self.lines):
            self.emit(1, "if False: yield 0")
        return "\n".join(self.lines)

    # ---------------------------------------------------------------- statements
    def block(self, toks, ind):
        i, n = 0, len(toks)
        while i < n:
            i = self.statement(toks, i, ind)

    def _match(self, toks, i, open_, close):
        depth, j = 0, i
        while True:
            depth += toks[j][1] == open_
            depth -= toks[j][1] == close
            j += 1
            if depth == 0:
                return j

    def _stmt_end(self, toks, i):
        """index after the statement starting at i (compound, control or simple)"""
        v = toks[i][1]
        if v == "{":
            return self._match(toks, i, "{", "}")
        if v in ("for", "while", "if"):
            j = self._match(toks, i + 1, "(", ")")
            j = self._stmt_end(toks, j)
            if v == "if" and j < len(toks) and toks[j][1] == "else":
                j = self._stmt_end(toks, j + 1)
            return j
        depth, j = 0, i
        while True:
            depth += toks[j][1] in "([{"
            depth -= toks[j][1] in ")]}"
            if toks[j][1] == ";" and depth == 0:
                return j + 1
            j += 1

    def _body(self, toks, i, ind):
        """translate the statement starting at i as a (possibly braced) body; returns the index after it"""
        j = self._stmt_end(toks, i)
        n0 = len(self.lines)
        if toks[i][1] == "{":
            self.block(toks[i + 1:j - 1], ind)
        else:
            self.block(toks[i:j], ind)
        if len(self.lines) == n0:
            self.emit(ind, "pass")
        return j

    def statement(self, toks, i, ind):
        kind, v = toks[i]
        if v == ";":
            return i + 1
        if v == "{":
            j = self._match(toks, i, "{", "}")
            self.block(toks[i + 1:j - 1], ind)
            return j
        if v == "if":
            j = self._match(toks, i + 1, "(", ")")
            cond = self.expr(toks[i + 2:j - 1], ind)
            self.emit(ind, "if %s:" % cond)
            j = self._body(toks, j, ind + 1)
            if j < len(toks) and toks[j][1] == "else":
                self.emit(ind, "else:")
                j = self._body(toks, j + 1, ind + 1)
            return j
        if v == "while":
            j = self._match(toks, i + 1, "(", ")")
            cond = self.expr(toks[i + 2:j - 1], ind, pre_ok=False)
            self.emit(ind, "while %s:" % cond)
            self.loop_incr.append(None)
            j = self._body(toks, j, ind + 1)
            self.loop_incr.pop()
            return j
        if v == "for":
            j = self._match(toks, i + 1, "(", ")")
            init, cond, incr = split_top(toks[i + 2:j - 1], ";")
            if init:
                self.simple(init, ind)
            self.emit(ind, "while %s:" % (self.expr(cond, ind, pre_ok=False) if cond else "True"))
            self.loop_incr.append(incr)
            j = self._body(toks, j, ind + 1)
            self.loop_incr.pop()
            for part in split_top(incr):
                if part:
                    self.simple(part, ind + 1)
            return j
        if v == "return":
            j = self._stmt_end(toks, i)
            e = toks[i + 1:j - 1]
            if e and typed() and any(w in self.tnames or w in ("T", "algType", "float", "double") for w in self.fd.ret):
                self.emit(ind, "return %s(%s)" % (self.caster(self.fd.ret), self.expr(e, ind)))
            else:
                self.emit(ind, "return %s" % self.expr(e, ind) if e else "return")
            return j
        if v == "continue":
            incr = self.loop_incr[-1]
            if incr:
                for part in split_top(incr):
                    if part:
                        self.simple(part, ind)
            self.emit(ind, "continue")
            return i + 2
        if v == "break":
            self.emit(ind, "break")
            return i + 2
        j = self._stmt_end(toks, i)
        self.simple(toks[i:j - 1], ind)
        return j

    def is_decl(self, toks):
        v = toks[0][1]
        if v in ("__shared__", "extern"):
            return True
        if v in TYPE_WORDS or (v in self.tnames and len(toks) > 1 and (toks[1][0] == "id" or toks[1][1] == "*")):
            return len(toks) > 1 and (toks[1][0] == "id" or toks[1][1] in ("*", "&")) and toks[1][1] != "("
        return False

    def simple(self, toks, ind):
        """declaration or expression statement (no trailing ;)"""
        if not toks:
            return
        if toks[0][1] in ("printf", "assert"):
            return
        if self.is_decl(toks):
            return self.declaration(toks, ind)
        # sync
        if toks[0][1] in ("__syncthreads", "hd__syncthreads"):
            if self.tu.device:
                self.emit(ind, "yield 1")
            return
        s = self.expr(toks, ind, stmt=True)
        if s:
            self.emit(ind, s)
            self.lockstep(ind)

    def split_declarators(self, toks):
        """comma split of a declarator list; the commas of an explicit template argument list (name<...>) belong to their expression"""
        parts, cur, depth, angle = [], [], 0, 0
        for x, t in enumerate(toks):
            v = t[1]
            if v in "([{":
                depth += 1
            elif v in ")]}":
                depth -= 1
            elif v == "<" and x > 0 and toks[x - 1][0] == "id" and (toks[x - 1][1] in self.tu.funcs or toks[x - 1][1] in ("static_cast", "reinterpret_cast", "shared_memory_proxy")):
                angle += 1
            elif v == ">" and angle:
                angle -= 1
            if v == "," and depth == 0 and angle == 0:
                parts.append(cur); cur = []
            else:
                cur.append(t)
        if cur or parts:
            parts.append(cur)
        return parts

    def lockstep(self, ind):
        """Device mode, inside a function that reaches a barrier: the threads of a block advance STATEMENT by statement (yield 0) between barriers (yield 1), like
        the lock step of a warp -- the reference has sections that are only correct under it (invHuu_dim4 reads the adjugate's first row into `val` in every
        thread and then overwrites it in place without a barrier in between, bpHelpers.cuh:169-181)."""
        if self.tu.device and self.fd.syncs:
            self.emit(ind, "yield 0")

    def declaration(self, toks, ind):
        shared = False
        i = 0
        while toks[i][1] in ("__shared__", "extern", "const", "static", "volatile"):
            shared |= toks[i][1] == "__shared__"
            i += 1
        # type words
        base = []
        while toks[i][0] == "id" and (toks[i][1] in TYPE_WORDS or toks[i][1] in self.tnames) and not (toks[i + 1][1] in ("=", ",", "[") or i + 1 >= len(toks)):
            base.append(toks[i][1]); i += 1
        is_int = any(b in ("int", "unsigned", "bool", "long", "size_t", "char") for b in base)
        for d in self.split_declarators(toks[i:]):
            stars = 0
            k = 0
            while d[k][1] in ("*", "&"):
                stars += 1; k += 1
            name = d[k][1]
            rest = d[k + 1:]
            if "timeval" in base:
                self.emit(ind, "%s = Struct(tv_sec=0, tv_usec=0)" % name)
                self.scalars.add(name)
                continue
            if "dim3" in base:
                args = []
                if rest and rest[0][1] == "(":
                    args = [self.expr(a, ind) for a in split_top(rest[1:-1])]
                self.emit(ind, "%s = Dim3(%s)" % (name, ", ".join(args)))
                continue
            if rest and rest[0][1] == "[":
                j = self._match(rest, 0, "[", "]")
                size = self.expr(rest[1:j - 1], ind)
                fill = ", 0" if is_int else ((", %s(0.0)" % self.caster(base)) if typed() else "")
                if shared:
                    self.emit(ind, "%s = _t.blk.shared(%r, %s%s)" % (name, self.fd.pyname + "." + name, size, fill if typed() else ""))
                else:
                    self.emit(ind, "%s = Ptr.alloc(%s%s)" % (name, size, fill))
                continue
            if rest and rest[0][1] == "=":
                val = self.expr(rest[1:], ind)
                if stars == 0 and not is_int and base and base[0] != "auto":
                    val = "%s(%s)" % (self.caster(base), val)
                    if typed(): self.vtypes[name] = self.caster(base)
                elif stars == 0 and is_int and "bool" not in base:
                    val = "int(%s)" % val
                mnull = re.match(r"^\(?([A-Za-z_]\w*) [+-] ", val) if stars else None
                if mnull:                                          # T *q = p + offset with p possibly null (an absent optional array): C forms the pointer, nobody dereferences it
                    self.emit(ind, "%s = None if %s is None else %s" % (name, mnull.group(1), val))
                else:
                    self.emit(ind, "%s = %s" % (name, val))
                self.lockstep(ind)
            else:
                if typed() and stars == 0 and not is_int and base and base[0] != "auto":
                    self.vtypes[name] = self.caster(base)
                    self.emit(ind, "%s = %s(0.0)" % (name, self.caster(base)))
                else:
                    self.emit(ind, "%s = %s" % (name, "None" if stars else ("0" if is_int else "0.0")))
            if stars == 0:
                self.scalars.add(name)

    # ---------------------------------------------------------------- expressions (precedence climbing -> Python text)
    def expr(self, toks, ind, stmt=False, pre_ok=True):
        self._toks, self._pos, self._ind, self._pre_ok = list(toks) + [("eof", "")], 0, ind, pre_ok
        self._post = []
        s = self.assignment(stmt)
        if self._toks[self._pos][0] != "eof":
            raise SyntaxError("trailing tokens in %s: %r" % (self.fd.name, [t[1] for t in self._toks[self._pos:self._pos + 8]]))
        if stmt:
            if re.match(r"^[A-Za-z_]\w*$", s or ""):
                s = ""
            for p in self._post:
                s = (s + "; " + p) if s else p
        elif self._post:                                            # out-arguments (&scalar) of a call inside a condition / initialiser: evaluate into a temporary first,
            assert pre_ok, "copy-back inside a loop condition: " + " ".join(t[1] for t in toks)
            tmp = self.newtmp()                                    # then copy the boxed scalars back, then use the value
            self.emit(ind, "%s = %s" % (tmp, s))
            for p in self._post:
                self.emit(ind, p)
            s = tmp
        return s

    def peek(self, k=0):
        return self._toks[self._pos + k][1]

    def nxt(self):
        t = self._toks[self._pos]; self._pos += 1
        return t

    def assignment(self, stmt):
        save = self._pos
        # comma-separated expression statements (e.g. "a = 1, b = 2") are not used by the reference; an assignment has an lvalue on the left
        lhs = self.ternary()
        op = self.peek()
        if op in ("=", "+=", "-=", "*=", "/=", "%=", "|=", "&=", "^=", "<<=", ">>="):
            self.nxt()
            rhs = self.assignment(False)
            if not stmt:
                raise SyntaxError("assignment inside an expression in %s" % self.fd.name)
            cv = self.vtypes.get(lhs) if typed() else None       # float32 mode: the value converts to the variable's type (array elements convert in Ptr.__setitem__)
            if cv:
                if op == "=":
                    return "%s = %s(%s)" % (lhs, cv, rhs)
                if op == "/=":
                    return "%s = %s(c_div(%s, %s))" % (lhs, cv, lhs, rhs)
                if op in ("+=", "-=", "*="):
                    return "%s = %s(%s %s (%s))" % (lhs, cv, lhs, op[0], rhs)
            if op == "/=":
                return "%s = c_div(%s, %s)" % (lhs, lhs, rhs)
            if op == "%=":
                return "%s = c_mod(%s, %s)" % (lhs, lhs, rhs)
            return "%s %s %s" % (lhs, op, rhs)
        if stmt and lhs in ("", "None"):
            return ""
        return lhs

    def ternary(self):
        c = self.binary(0)
        if self.peek() == "?":
            self.nxt()
            a = self.assignment(False)
            assert self.nxt()[1] == ":"
            b = self.ternary()
            return "(%s if %s else %s)" % (a, c, b)
        return c

    BIN = [["||"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="], ["<", ">", "<=", ">="], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]
    PYOP = {"||": "or", "&&": "and"}

    def binary(self, level):
        if level == len(self.BIN):
            return self.unary()
        lhs = self.binary(level + 1)
        while self.peek() in self.BIN[level]:
            op = self.nxt()[1]
            rhs = self.binary(level + 1)
            if op == "/":
                lhs = "c_div(%s, %s)" % (lhs, rhs)
            elif op == "%":
                lhs = "c_mod(%s, %s)" % (lhs, rhs)
            elif op in ("==", "!=") and (rhs == "None" or lhs == "None"):
                other = lhs if rhs == "None" else rhs
                lhs = "(%s is%s None)" % (other, "" if op == "==" else " not")
            else:
                lhs = "(%s %s %s)" % (lhs, self.PYOP.get(op, op), rhs)
        return lhs

    def unary(self):
        v = self.peek()
        if v == "!":
            self.nxt(); return "(not %s)" % self.unary()
        if v == "-":
            self.nxt(); return "(-%s)" % self.unary()
        if v == "+":
            self.nxt(); return self.unary()
        if v == "~":
            self.nxt(); return "(~%s)" % self.unary()
        if v == "*":
            self.nxt(); return "%s[0]" % self.unary()
        if v == "&":
            self.nxt()
            inner = self.unary()
            if inner.endswith("]"):                                # &base[index] -> base + index (the bracket that closes last)
                depth = 0
                for k in range(len(inner) - 1, -1, -1):
                    depth += inner[k] == "]"
                    depth -= inner[k] == "["
                    if depth == 0:
                        return "padd(%s, %s)" % (inner[:k], inner[k + 1:-1])
            if inner in self.scalars:                              # &scalar: an out-argument -- box it, copy back after the statement
                box = self.newtmp()
                self.emit(self._ind, "%s = [%s]" % (box, inner))
                self._post.append("%s = %s[0]" % (inner, box))
                return box
            return inner                                            # &array / &function: the thing itself
        if v in ("++", "--"):
            self.nxt()
            tgt = self.unary()
            self._post.insert(0, "%s %s= 1" % (tgt, v[0]))          # statement-level only (expr() asserts)
            return ""
        if v == "(":
            # cast: (T) x, (int) x, (double) x, (T *) x
            if self._toks[self._pos + 1][0] == "id" and (self.peek(1) in TYPE_WORDS or self.peek(1) in self.tnames):
                j = self._pos + 1
                words = []
                while self._toks[j][0] == "id" and (self._toks[j][1] in TYPE_WORDS or self._toks[j][1] in self.tnames):
                    words.append(self._toks[j][1]); j += 1
                stars = 0
                while self._toks[j][1] == "*":
                    stars += 1; j += 1
                if self._toks[j][1] == ")":
                    self._pos = j + 1
                    inner = self.unary()
                    if stars:
                        return inner
                    if any(w in ("int", "unsigned", "long", "size_t") for w in words):
                        return "int(%s)" % inner
                    return "%s(%s)" % (self.caster(words), inner)
        return self.postfix()

    @staticmethod
    def _balanced(s):
        d = 0
        for ch in s:
            d += ch in "([{"
            d -= ch in ")]}"
            if d < 0:
                return False
        return d == 0

    def args(self):
        """after '(' consumed: list of translated arguments, consumes ')'"""
        out = []
        if self.peek() == ")":
            self.nxt(); return out
        while True:
            out.append(self.assignment(False))
            v = self.nxt()[1]
            if v == ")":
                return out
            assert v == ",", v

    def postfix(self):
        kind, v = self.nxt()
        if kind == "num":
            s = re.sub(r"[uUlL]+$", "", v)
            if re.search(r"[.eE]", s) and not s.lower().startswith("0x"):
                if typed() and re.search(r"[fF]$", s):
                    s = "F32(%s)" % re.sub(r"[fF]$", "", s)           # a float literal
                else:
                    s = re.sub(r"[fF]$", "", s)
            base = s
        elif kind == "id":
            base = self.primary_id(v)
        elif v == "(":
            inner = self.assignment(False)
            assert self.nxt()[1] == ")"
            base = "(%s)" % inner
        else:
            raise SyntaxError("unexpected token %r in %s" % (v, self.fd.name))
        while True:
            v = self.peek()
            if v == "[":
                self.nxt()
                idx = self.assignment(False)
                assert self.nxt()[1] == "]"
                base = "%s[%s]" % (base, idx)
            elif v in (".", "->"):
                self.nxt()
                member = self.nxt()[1]
                if self.peek() == "(":                              # a method of a host object: std::vector (push_back / back), std::mutex (lock / unlock)
                    self.nxt()
                    a = self.args()
                    if member == "push_back":
                        base = "%s.append(%s)" % (base, ", ".join(a))
                    elif member == "back":
                        base = "%s[-1]" % base
                    else:
                        base = "%s.%s(%s)" % (base, member, ", ".join(a))
                else:
                    base = "%s.%s" % (base, member)
            elif v in ("++", "--"):
                self.nxt()
                self._post.append("%s %s= 1" % (base, v[0]))
                # value of a post-increment is only used as a statement in the reference
            else:
                return base

    def primary_id(self, v):
        if v == "std" and self.peek() == "::":                      # std::floor, std::ceil: the C functions of the same name
            self.nxt()
            v = self.nxt()[1]
        if v in ("nullptr", "NULL"):
            return "None"
        if v == "true":
            return "True"
        if v == "false":
            return "False"
        if v in ("threadIdx", "blockIdx", "blockDim", "gridDim") and self.peek() == ".":
            self.nxt()
            return CUDA_VARS[(v, self.nxt()[1])]
        if v == "static_cast" or v == "reinterpret_cast":
            assert self.nxt()[1] == "<"
            words = []
            while self.peek() != ">":
                words.append(self.nxt()[1])
            self.nxt()
            assert self.nxt()[1] == "("
            inner = self.assignment(False)
            assert self.nxt()[1] == ")"
            if "*" in words:
                return inner
            if any(w in ("int", "unsigned", "long") for w in words):
                return "int(%s)" % inner
            return "%s(%s)" % (self.caster(words), inner)
        if v == "sizeof":
            assert self.nxt()[1] == "("
            while self.nxt()[1] != ")":
                pass
            return str(SIZEOF)
        if v == "shared_memory_proxy":
            while self.peek() != "(":
                self.nxt()
            self.nxt(); assert self.nxt()[1] == ")"
            return "_t.blk.extern"
        # calls
        targs = None
        if v in self.tu.funcs and self.peek() == "<":
            # explicit template arguments
            j = self._pos
            depth = 0
            while True:
                if self._toks[j][0] == "eof":
                    raise SyntaxError("unterminated template argument list after %s in %s" % (v, self.fd.name))
                depth += self._toks[j][1] == "<"
                depth -= self._toks[j][1] == ">"
                j += 1
                if depth == 0:
                    break
            inner = self._toks[self._pos + 1:j - 1]
            sub = []
            for part in split_template_args(inner):
                if len(part) == 1 and (part[0][1] in self.tnames and part[0][1] == "T" or part[0][1] in ("float", "double", "T")):
                    sub.append(self.caster([part[0][1]]))
                else:
                    ft = FuncTranslator(self.tu, self.fd)
                    ft.scalars, ft.tnames = self.scalars, self.tnames
                    sub.append(ft.expr(part, self._ind, pre_ok=False))
            targs = sub
            self._pos = j
        if self.peek() == "<<<":                                    # kernel<<<grid, block, shared bytes, stream>>>(args)
            self.nxt()
            cfg = []
            while True:
                cfg.append(self.assignment(False))
                t = self.nxt()[1]
                if t == ">>>":
                    break
                assert t == ",", t
            assert self.nxt()[1] == "("
            a = self.args()
            return "launch_kernel(%r, (%s), [%s], [%s])" % (v, "".join(t + ", " for t in (targs or [])), ", ".join(cfg), ", ".join(a))
        if self.peek() == "(" and v in CUDA_RT and v not in self.tu.funcs:
            self.nxt()
            return "%s(%s)" % (CUDA_RT[v], ", ".join(self.args()))
        if self.peek() == "(" and (v in self.tu.funcs or v in MATH or v in ("__syncthreads", "hd__syncthreads")):
            self.nxt()
            a = self.args()
            if v in ("__syncthreads", "hd__syncthreads"):
                return ""
            if v in self.tu.funcs:
                fd = self.tu.overload(v, len(a))
                self.tu.require(fd)
                call = "%s(_t, (%s)%s)" % (fd.pyname, "".join(t + ", " for t in (targs or [])), "".join(", " + x for x in a))
                if self.tu.device and fd.syncs:
                    if not self._pre_ok:
                        raise SyntaxError("call of a barrier-reaching function in a loop condition: %s in %s" % (v, self.fd.name))
                    tmp = self.newtmp()
                    self.emit(self._ind, "%s = yield from %s" % (tmp, call))
                    return tmp
                return call
            return "%s(%s)" % (MATH[v], ", ".join(a))
        return v


# ------------------------------------------------------------------------------------------------------------------ SIMT emulation
def launch(ns, kernel, grid, block, args, tp=(), extern_elems=0):
    """kernel<<<grid, block, extern_elems * sizeof(T)>>>(args): every block's threads advanced barrier to barrier"""
    gx, gy = (grid.x, grid.y) if isinstance(grid, Dim3) else ((grid, 1) if isinstance(grid, int) else grid)
    bx, by = (block.x, block.y) if isinstance(block, Dim3) else ((block, 1) if isinstance(block, int) else block)
    fn = ns[kernel]
    for biy in range(gy):
        for bix in range(gx):
            blk = Block(extern_elems)
            gens = []
            for tiy in range(by):
                for tix in range(bx):
                    g = fn(Thread(tix, tiy, bx, by, bix, biy, gx, gy, blk), tp, *args)
                    gens.append(g)
            if not hasattr(gens[0], "__next__"):                 # a kernel without barriers is a plain function: already executed thread after thread
                continue
            # lock step between barriers: every live thread that is not waiting at a barrier executes one statement per round; a barrier releases when every
            # live thread has arrived (threads that returned no longer take part, as on the hardware)
            live, waiting = list(gens), set()
            while live:
                progressed = False
                for g in list(live):
                    if g in waiting:
                        continue
                    try:
                        if next(g) == 1:
                            waiting.add(g)
                        progressed = True
                    except StopIteration:
                        live.remove(g)
                        progressed = True
                if live and all(g in waiting for g in live):
                    waiting.clear()
                elif not progressed:
                    raise RuntimeError("deadlock in the emulated block")


def load(root, entry, names, predefined=None, device=False, skip_includes=("threadUtils.h", "exampleUtils.cuh", "DDPWrappers.cuh", "MPCHelpers.cuh", "LCMHelpers.cuh")):
    """preprocess `entry` (with its quoted includes) under the given -D definitions, translate `names` and everything they call.
    Returns (namespace, preprocessor, translator)."""
    pp = Preprocessor(root, predefined, cuda_arch=device, skip_includes=skip_includes)
    pp.process_file(entry)
    tu = Translator(pp.out, device)
    ns = tu.build(names)
    return ns, pp, tu
