"""Reading MANO model files as data: nothing a file names is ever imported, resolved or called.

The official MANO_RIGHT.pkl / MANO_LEFT.pkl are Python pickles that name chumpy, scipy and numpy callables; the
reference loads them with pickle.load (mano/model.py:151-153), which imports and runs whatever the file names.  Here
the `pickle` machinery is not used at all.  The byte stream is tokenised (`pickletools.genops` only splits it into
opcodes and their literal arguments) and a small stack machine assembles plain containers; a global named by the file
stays an opaque (module, name) tag and a "call" of it is only RECORDED, as the tag with its literal arguments and
state.  Afterwards the recorded tree is interpreted by this module's own code, which knows the few layouts a MANO file
uses -- a numpy array (shape, dtype string, raw bytes), a scipy CSC/CSR matrix (its three arrays and shape), the two
chumpy wrappers (their array payload), a set -- and refuses every other tag.  `.npz` files are read with
allow_pickle=False.
"""
from __future__ import annotations

import pickletools

import numpy as np

FIELDS = ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "hands_components", "hands_mean",
          "kintree_table", "f")


class ModelFileError(ValueError):
    """The file is not a MANO model file this reader understands (or names something that is not plain data)."""


class Struct(object):
    def __init__(self, **kwargs):
        for key, val in kwargs.items():
            setattr(self, key, val)


def to_np(array, dtype=np.float32):
    if 'scipy.sparse' in str(type(array)):
        array = np.array(array.todense(), dtype=dtype)
    elif hasattr(array, 'detach'):
        array = array.detach().cpu().numpy()
    return np.array(array, dtype=dtype)


# ------------------------------------------------------------------ phase 1: tokens -> recorded tree
class _Tag:
    """A global the file names.  Never looked up: it only labels a record."""
    __slots__ = ("module", "name")

    def __init__(self, module, name):
        self.module, self.name = module, name

    def key(self):
        return (self.module, self.name)


class _Record:
    """'tag(*args)' followed by optional state, exactly as written in the file; nothing has been called."""
    __slots__ = ("tag", "args", "state")

    def __init__(self, tag, args):
        self.tag, self.args, self.state = tag, args, None


_MARK = object()
_MAX_OPS = 2_000_000


def _tokens_to_tree(data: bytes):
    stack, memo = [], {}

    def pop_to_mark():
        for i in range(len(stack) - 1, -1, -1):
            if stack[i] is _MARK:
                items = stack[i + 1:]
                del stack[i:]
                return items
        raise ModelFileError("malformed model file (no mark)")

    n_ops = 0
    for op, arg, _pos in pickletools.genops(data):
        n_ops += 1
        if n_ops > _MAX_OPS:
            raise ModelFileError("model file too long")
        name = op.name
        if name in ("PROTO", "FRAME"):
            continue
        if name == "STOP":
            if len(stack) != 1:
                raise ModelFileError("malformed model file (stack at end)")
            return stack[0]
        if name == "MARK":
            stack.append(_MARK)
        elif name in ("NONE",):
            stack.append(None)
        elif name == "NEWTRUE":
            stack.append(True)
        elif name == "NEWFALSE":
            stack.append(False)
        elif name in ("BININT", "BININT1", "BININT2", "LONG1", "LONG4", "INT", "LONG", "BINFLOAT", "FLOAT"):
            stack.append(arg)
        elif name in ("SHORT_BINSTRING", "BINSTRING", "STRING"):
            # a Python-2 byte string; the tokeniser hands it over latin-1 decoded
            stack.append(arg.encode("latin-1") if isinstance(arg, str) else bytes(arg))
        elif name in ("BINUNICODE", "SHORT_BINUNICODE", "BINUNICODE8", "UNICODE"):
            stack.append(arg)
        elif name in ("BINBYTES", "SHORT_BINBYTES", "BINBYTES8"):
            stack.append(bytes(arg))
        elif name == "EMPTY_DICT":
            stack.append({})
        elif name == "EMPTY_LIST":
            stack.append([])
        elif name == "EMPTY_TUPLE":
            stack.append(())
        elif name == "EMPTY_SET":
            stack.append(set())
        elif name == "TUPLE":
            stack.append(tuple(pop_to_mark()))
        elif name == "TUPLE1":
            stack[-1:] = [tuple(stack[-1:])]
        elif name == "TUPLE2":
            stack[-2:] = [tuple(stack[-2:])]
        elif name == "TUPLE3":
            stack[-3:] = [tuple(stack[-3:])]
        elif name == "LIST":
            stack.append(list(pop_to_mark()))
        elif name == "DICT":
            items = pop_to_mark()
            stack.append(dict(zip(items[0::2], items[1::2])))
        elif name == "APPEND":
            v = stack.pop()
            stack[-1].append(v)
        elif name == "APPENDS":
            items = pop_to_mark()
            stack[-1].extend(items)
        elif name == "ADDITEMS":
            items = pop_to_mark()
            stack[-1].update(items)
        elif name == "SETITEM":
            v = stack.pop()
            k = stack.pop()
            _dict_of(stack[-1])[_hashable(k)] = v
        elif name == "SETITEMS":
            items = pop_to_mark()
            d = _dict_of(stack[-1])
            for k, v in zip(items[0::2], items[1::2]):
                d[_hashable(k)] = v
        elif name in ("BINPUT", "LONG_BINPUT", "PUT"):
            memo[arg] = stack[-1]
        elif name == "MEMOIZE":
            memo[len(memo)] = stack[-1]
        elif name in ("BINGET", "LONG_BINGET", "GET"):
            if arg not in memo:
                raise ModelFileError("malformed model file (memo)")
            stack.append(memo[arg])
        elif name == "GLOBAL":
            module, _, gname = arg.partition(" ")
            stack.append(_Tag(module, gname))
        elif name == "STACK_GLOBAL":
            gname = stack.pop()
            module = stack.pop()
            if not (isinstance(module, str) and isinstance(gname, str)):
                raise ModelFileError("malformed model file (global name)")
            stack.append(_Tag(module, gname))
        elif name == "REDUCE":
            args = stack.pop()
            tag = stack.pop()
            stack.append(_record(tag, args))
        elif name == "NEWOBJ":
            args = stack.pop()
            tag = stack.pop()
            stack.append(_record(tag, args))
        elif name == "BUILD":
            state = stack.pop()
            rec = stack[-1]
            if not isinstance(rec, _Record):
                raise ModelFileError("model file sets state on something that is not a record")
            rec.state = state
        else:
            raise ModelFileError(f"model file uses pickle opcode {name}, which plain array data does not need")
    raise ModelFileError("malformed model file (no end)")


def _record(tag, args):
    if not isinstance(tag, _Tag) or not isinstance(args, tuple):
        raise ModelFileError("malformed model file (call of a non-global)")
    return _Record(tag, args)


def _dict_of(obj):
    if not isinstance(obj, dict):
        raise ModelFileError("malformed model file (item assignment to a non-dict)")
    return obj


def _hashable(k):
    if isinstance(k, (str, bytes, int, float, bool, tuple)) or k is None:
        return k.decode("latin-1") if isinstance(k, bytes) else k
    raise ModelFileError("malformed model file (unhashable key)")


# ------------------------------------------------------------------ phase 2: recorded tree -> arrays
_NDARRAY_MAKERS = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct")}
_SPARSE = {("scipy.sparse.csc", "csc_matrix"): "csc", ("scipy.sparse._csc", "csc_matrix"): "csc",
           ("scipy.sparse.csr", "csr_matrix"): "csr", ("scipy.sparse._csr", "csr_matrix"): "csr"}
_CHUMPY = {("chumpy.ch", "Ch"), ("chumpy.reordering", "Select")}
_OBJECT_MAKERS = {("copy_reg", "_reconstructor"), ("copyreg", "_reconstructor")}
_PLAIN_DTYPE = set("?bhilqBHILQefdg") | {f"{c}{n}" for c in "iufb" for n in (1, 2, 4, 8)}


def _text(v):
    return v.decode("latin-1") if isinstance(v, bytes) else v


def _dtype(rec):
    """numpy.dtype('<code>', ...) + its state tuple (version, byte order, ...): plain numeric types only."""
    if not (isinstance(rec, _Record) and rec.tag.key() == ("numpy", "dtype")):
        raise ModelFileError("array without a plain dtype")
    code = _text(rec.args[0])
    if not isinstance(code, str) or code.lstrip("<>=|") not in _PLAIN_DTYPE:
        raise ModelFileError(f"array dtype {code!r} is not a plain numeric type")
    dt = np.dtype(code)
    if isinstance(rec.state, tuple) and len(rec.state) >= 2:
        order = _text(rec.state[1])
        if any(s is not None for s in rec.state[2:5]):
            raise ModelFileError("structured array dtypes are not model data")
        if order in ("<", ">"):
            dt = dt.newbyteorder(order)
    return dt


def _raw_bytes(v):
    """An array's buffer: a byte string, or -- how Python 3 writes bytes at protocol 2 -- the record
    `_codecs.encode(text, 'latin1')`, undone here."""
    if isinstance(v, bytes):
        return v
    if isinstance(v, _Record) and v.tag.key() == ("_codecs", "encode") and len(v.args) == 2 \
            and isinstance(v.args[0], str) and _text(v.args[1]).replace("-", "").lower() in ("latin1", "iso88591"):
        return v.args[0].encode("latin-1")
    raise ModelFileError("array without a raw buffer")


def _ndarray(rec):
    st = rec.state
    if not (isinstance(st, tuple) and len(st) == 5):
        raise ModelFileError("array record without (version, shape, dtype, order, buffer)")
    _version, shape, dt, fortran, raw = st
    if not (isinstance(shape, tuple) and all(isinstance(s, int) and s >= 0 for s in shape)):
        raise ModelFileError("array shape is not a tuple of sizes")
    dt = _dtype(dt)
    raw = _raw_bytes(raw)
    count = int(np.prod(shape, dtype=np.int64)) if shape else 1
    if len(raw) != count * dt.itemsize:
        raise ModelFileError("array buffer does not match its shape")
    a = np.frombuffer(raw, dtype=dt, count=count)
    a = a.reshape(shape, order="F" if fortran else "C")
    return a.astype(dt.newbyteorder("="), copy=True)


def _scalar(rec):
    dt = _dtype(rec.args[0])
    raw = _raw_bytes(rec.args[1])
    if len(raw) != dt.itemsize:
        raise ModelFileError("scalar buffer does not match its type")
    return np.frombuffer(raw, dtype=dt, count=1)[0]


def _object_state(rec):
    """State dict of an object written through copy_reg._reconstructor(cls, object, None) or NEWOBJ."""
    st = rec.state
    if st is None:
        return {}
    if isinstance(st, tuple) and len(st) == 2 and isinstance(st[0], dict) and st[1] is None:
        st = st[0]
    if not isinstance(st, dict):
        raise ModelFileError("object state is not a dict")
    return {_text(k): v for k, v in st.items()}


def _chumpy_value(kind, state):
    if kind == "Ch":
        if "x" not in state:
            raise ModelFileError("chumpy array without a value")
        return np.asarray(_interpret(state["x"]))
    a = np.asarray(_interpret(state.get("a")))
    idxs = np.asarray(_interpret(state.get("idxs")))
    if not np.issubdtype(idxs.dtype, np.integer):
        raise ModelFileError("chumpy selection with non-integer indices")
    out = a.ravel()[idxs]
    shape = _interpret(state.get("preferred_shape"))
    return out.reshape(shape) if shape is not None else out


def _sparse(fmt, state):
    import scipy.sparse as sp
    data, indices, indptr = (np.asarray(_interpret(state[k])) for k in ("data", "indices", "indptr"))
    shape = _interpret(state.get("_shape", state.get("shape")))
    if not (isinstance(shape, tuple) and len(shape) == 2):
        raise ModelFileError("sparse matrix without a 2-d shape")
    cls = sp.csc_matrix if fmt == "csc" else sp.csr_matrix
    return cls((data, indices, indptr), shape=tuple(int(s) for s in shape))


def _interpret(node, depth=0):
    if depth > 64:
        raise ModelFileError("model file nests too deeply")
    if node is None or isinstance(node, (bool, int, float, str, bytes)):
        return node
    if isinstance(node, dict):
        return {_text(k): _interpret(v, depth + 1) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        out = [_interpret(v, depth + 1) for v in node]
        return out if isinstance(node, list) else tuple(out)
    if isinstance(node, set):
        return {_interpret(v, depth + 1) for v in node}
    if isinstance(node, _Tag):
        raise ModelFileError(f"model file names {node.module}.{node.name}, which is not plain array data; refusing it")
    if isinstance(node, _Record):
        key = node.tag.key()
        if key in _NDARRAY_MAKERS:
            return _ndarray(node)
        if key in (("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")):
            return _scalar(node)
        if key == ("numpy", "dtype"):
            return _dtype(node)
        if key in (("__builtin__", "set"), ("builtins", "set")):
            return set(_interpret(node.args[0], depth + 1)) if node.args else set()
        target = node.tag
        if key in _OBJECT_MAKERS:          # _reconstructor(cls, object, None): the class is the first argument
            if not (node.args and isinstance(node.args[0], _Tag)):
                raise ModelFileError("object record without a class")
            target = node.args[0]
        tkey = target.key()
        if tkey in _CHUMPY:
            return _chumpy_value(target.name, _object_state(node))
        if tkey in _SPARSE:
            return _sparse(_SPARSE[tkey], _object_state(node))
        raise ModelFileError(f"model file names {target.module}.{target.name}, which is not plain array data; "
                             f"refusing it")
    raise ModelFileError("unexpected value in model file")


def read_pickled_arrays(path: str):
    """The containers and arrays a pickle file holds, read as data (see the module docstring)."""
    with open(path, "rb") as fh:
        data = fh.read()
    try:
        return _interpret(_tokens_to_tree(data))
    except ModelFileError:
        raise
    except RecursionError:
        raise ModelFileError("model file nests too deeply") from None
    except Exception as e:                 # a truncated or corrupt stream: the tokeniser's own errors, index/type/value errors
        raise ModelFileError(f"malformed model file: {type(e).__name__}: {e}") from None


def read_model_file(path: str) -> dict:
    """dict with the FIELDS the MANO layer needs, from a .pkl or .npz model file."""
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as f:
            return {k: f[k] for k in f.files}
    data = read_pickled_arrays(path)
    if not isinstance(data, dict):
        raise ModelFileError("MANO model file does not hold a dict")
    missing = [k for k in FIELDS if k not in data]
    if missing:
        raise ModelFileError(f"MANO model file lacks {missing}")
    return data
