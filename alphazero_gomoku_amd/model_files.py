"""Reading the reference's pickled model files without unpickling them.

The reference's Theano and numpy nets keep their weights as `pickle.dump(net_params, f, protocol=2)`
of a list of float arrays (policy_value_net.py:113-121, loaded by policy_value_net_numpy.py:87-90 with
`pickle.load(..., encoding='bytes')`).  A pickle is a program for a stack machine that can import and
call anything, so `pickle.load` is not used here.  `read_pickled_arrays` only DISASSEMBLES the file
(`pickletools.genops` yields opcodes and their constants; it imports nothing, calls nothing and
builds no objects) and replays the few plain-data opcodes a list of numpy arrays consists of on a
stack of inert Python values.  GLOBAL / STACK_GLOBAL and REDUCE leave records holding NAMES; the
names are compared with numpy's array-reconstruction entry points afterwards (and `_codecs encode`
with 'latin1', the way Python 3 spells a byte string below protocol 3) and the arrays are built here
from dtype string, shape and raw bytes.  Every other opcode, name or shape of data is
refused with ValueError."""
import pickletools

import numpy as np

_RECONSTRUCT = {"numpy.core.multiarray _reconstruct", "numpy._core.multiarray _reconstruct"}
_NDARRAY = {"numpy ndarray"}
_DTYPE = {"numpy dtype"}
_FLOATS = {"f4": "<f4", "f8": "<f8"}


class _Record:
    """What a REDUCE leaves behind: the (inert) callable description, its arguments, a BUILD state."""
    __slots__ = ("fn", "args", "state")

    def __init__(self, fn, args):
        self.fn, self.args, self.state = fn, args, None


class _Global:
    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name


def _as_bytes(v):
    if isinstance(v, str):
        return v.encode("latin-1")          # Python-2 str constants: raw bytes, one code point per byte
    return bytes(v)


def _as_text(v):
    return v.decode("latin-1") if isinstance(v, (bytes, bytearray)) else str(v)


def read_pickled_arrays(path, max_bytes=1 << 28):
    """The list of float arrays in a model file of the reference's Theano / numpy nets, as float32
    numpy arrays (pickle protocols 0-4 as numpy writes them under Python 2 or 3).  Nothing from the
    file is executed; see the module docstring.  ValueError when the file holds anything else."""
    with open(path, "rb") as f:
        data = f.read(max_bytes + 1)
    if len(data) > max_bytes:
        raise ValueError("%s: larger than %d bytes, not a model file of this net" % (path, max_bytes))
    mark = object()
    stack, memo = [], {}

    def pop_mark():
        for i in range(len(stack) - 1, -1, -1):
            if stack[i] is mark:
                items = stack[i + 1:]
                del stack[i:]
                return items
        raise ValueError("%s: MARK missing" % path)

    def need(k):
        if len(stack) < k:
            raise ValueError("%s: stack underflow" % path)

    try:
        ops = list(pickletools.genops(data))
    except Exception as e:                       # truncated or not a pickle at all
        raise ValueError("%s: not a pickle (%s)" % (path, e))
    for op, arg, _pos in ops:
        name = op.name
        if name in ("PROTO", "FRAME"):
            pass
        elif name == "EMPTY_LIST":
            stack.append([])
        elif name == "EMPTY_TUPLE":
            stack.append(())
        elif name == "MARK":
            stack.append(mark)
        elif name in ("BINPUT", "LONG_BINPUT", "PUT"):
            need(1)
            memo[int(arg)] = stack[-1]
        elif name == "MEMOIZE":
            need(1)
            memo[len(memo)] = stack[-1]
        elif name in ("BINGET", "LONG_BINGET", "GET"):
            if int(arg) not in memo:
                raise ValueError("%s: unknown memo entry" % path)
            stack.append(memo[int(arg)])
        elif name == "GLOBAL":
            stack.append(_Global(str(arg)))
        elif name == "STACK_GLOBAL":
            need(2)
            attr, mod = stack.pop(), stack.pop()
            stack.append(_Global("%s %s" % (_as_text(mod), _as_text(attr))))
        elif name in ("BININT", "BININT1", "BININT2", "INT", "LONG", "LONG1", "LONG4"):
            stack.append(arg if isinstance(arg, bool) else int(arg))
        elif name in ("SHORT_BINSTRING", "BINSTRING", "STRING", "BINBYTES", "SHORT_BINBYTES", "BINBYTES8"):
            stack.append(_as_bytes(arg))
        elif name in ("BINUNICODE", "SHORT_BINUNICODE", "BINUNICODE8", "UNICODE"):
            stack.append(str(arg))
        elif name == "NONE":
            stack.append(None)
        elif name in ("NEWFALSE", "NEWTRUE"):
            stack.append(name == "NEWTRUE")
        elif name in ("TUPLE1", "TUPLE2", "TUPLE3"):
            k = int(name[-1])
            need(k)
            items = tuple(stack[-k:])
            del stack[-k:]
            stack.append(items)
        elif name == "TUPLE":
            stack.append(tuple(pop_mark()))
        elif name == "LIST":
            stack.append(list(pop_mark()))
        elif name == "REDUCE":
            need(2)
            args, fn = stack.pop(), stack.pop()
            if not isinstance(fn, _Global) or not isinstance(args, tuple):
                raise ValueError("%s: REDUCE on something that is not a named global" % path)
            stack.append(_Record(fn.name, args))
        elif name == "BUILD":
            need(2)
            state = stack.pop()
            if not isinstance(stack[-1], _Record):
                raise ValueError("%s: BUILD on plain data" % path)
            stack[-1].state = state
        elif name == "APPENDS":
            items = pop_mark()
            need(1)
            if not isinstance(stack[-1], list):
                raise ValueError("%s: APPENDS on something that is not a list" % path)
            stack[-1].extend(items)
        elif name == "APPEND":
            need(2)
            item = stack.pop()
            if not isinstance(stack[-1], list):
                raise ValueError("%s: APPEND on something that is not a list" % path)
            stack[-1].append(item)
        elif name == "STOP":
            break
        else:
            raise ValueError("%s: opcode %s is not a plain-data opcode, file refused" % (path, name))
    if len(stack) != 1 or not isinstance(stack[0], list):
        raise ValueError("%s: the file does not hold one list" % path)
    return [_array_of(rec, path) for rec in stack[0]]


def _array_of(rec, path):
    """numpy's ndarray.__reduce__: _reconstruct(ndarray, (0,), b'b') + state (version, shape, dtype,
    is_fortran, raw bytes), the dtype itself being dtype('f4', False/0, True/1) + state (3, '<', ...)."""
    bad = ValueError("%s: a list entry is not a plain float array" % path)
    if not isinstance(rec, _Record) or rec.fn not in _RECONSTRUCT or not isinstance(rec.state, tuple) or len(rec.state) != 5:
        raise bad
    if not (len(rec.args) == 3 and isinstance(rec.args[0], _Global) and rec.args[0].name in _NDARRAY):
        raise bad
    version, shape, dt, fortran, raw = rec.state
    if not (isinstance(dt, _Record) and dt.fn in _DTYPE and len(dt.args) >= 1 and isinstance(dt.state, tuple) and len(dt.state) >= 2):
        raise bad
    code, order = _as_text(dt.args[0]), _as_text(dt.state[1])
    if code not in _FLOATS or order not in ("<", "=", "|"):
        raise ValueError("%s: array of dtype %s%s, expected little-endian float32/float64" % (path, order, code))
    if version != 1 or not isinstance(shape, tuple) or not all(isinstance(d, int) and not isinstance(d, bool) and d >= 0 for d in shape):
        raise bad
    if isinstance(raw, _Record) and raw.fn == "_codecs encode" and len(raw.args) == 2 and isinstance(raw.args[0], str) \
            and _as_text(raw.args[1]).lower().replace("-", "") in ("latin1", "iso88591"):
        raw = raw.args[0].encode("latin-1")      # how Python 3 writes a bytes object below protocol 3
    if not isinstance(raw, (bytes, bytearray)):
        raise bad
    a = np.frombuffer(bytes(raw), dtype=_FLOATS[code])
    if a.size != int(np.prod(shape, dtype=np.int64)):
        raise bad
    a = a.reshape(shape, order="F" if fortran else "C")
    return np.ascontiguousarray(a, dtype=np.float32)


def looks_like_pickle(path):
    """True for files that start like a binary pickle (protocol >= 2) or a protocol-0 list."""
    with open(path, "rb") as f:
        head = f.read(2)
    return head[:1] == b"\x80" or head in (b"(l", b"]q")
