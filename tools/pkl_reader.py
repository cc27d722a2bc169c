"""Read the *data* of a pickle file without executing anything from it.

The reference's result records (benchmark_data/*.pkl, written by its
utils.py:197-233) are pickles of a dict of floats, ints, strings, dicts and
numpy arrays.  ``pickle.load`` would import and call whatever the file names;
this reader does not: it walks the opcode stream with ``pickletools.genops``
(a disassembler) on a tiny stack machine of its own in which every GLOBAL is a
*symbol* (a string, never imported) and every REDUCE/BUILD is interpreted only
for the three numpy constructs these files contain
(``_reconstruct``+BUILD = ndarray, ``dtype``, ``scalar``), decoded with
``numpy.frombuffer`` from the raw bytes.  Any other global or opcode raises.
"""
from __future__ import annotations

import pickletools

import numpy as np


class _Sym:
    def __init__(self, module, name):
        self.module, self.name = module, name

    def __repr__(self):
        return f"<sym {self.module}.{self.name}>"


class _Dtype:
    def __init__(self, code):
        self.code = code
        self.byteorder = '='

    def np(self):
        dt = np.dtype(self.code)
        if self.byteorder in '<>':
            dt = dt.newbyteorder(self.byteorder)
        return dt


class _PendingArray:
    pass


_MARK = object()


def load_data(path):
    stack, memo = [], {}

    def pop_to_mark():
        items = []
        while True:
            x = stack.pop()
            if x is _MARK:
                break
            items.append(x)
        return items[::-1]

    with open(path, 'rb') as f:
        data = f.read()
    for op, arg, _pos in pickletools.genops(data):
        nm = op.name
        if nm in ('PROTO', 'FRAME'):
            continue
        elif nm == 'STOP':
            return stack.pop()
        elif nm == 'MARK':
            stack.append(_MARK)
        elif nm in ('EMPTY_DICT',):
            stack.append({})
        elif nm in ('EMPTY_LIST',):
            stack.append([])
        elif nm in ('EMPTY_TUPLE',):
            stack.append(())
        elif nm in ('MEMOIZE',):
            memo[len(memo)] = stack[-1]
        elif nm in ('BINPUT', 'LONG_BINPUT', 'PUT'):
            memo[int(arg)] = stack[-1]
        elif nm in ('BINGET', 'LONG_BINGET', 'GET'):
            stack.append(memo[int(arg)])
        elif nm in ('SHORT_BINUNICODE', 'BINUNICODE', 'BINUNICODE8', 'UNICODE',
                    'BINFLOAT', 'FLOAT', 'BININT', 'BININT1', 'BININT2', 'INT', 'LONG1', 'LONG4',
                    'SHORT_BINBYTES', 'BINBYTES', 'BINBYTES8', 'SHORT_BINSTRING', 'BINSTRING'):
            stack.append(arg)
        elif nm == 'NONE':
            stack.append(None)
        elif nm == 'NEWTRUE':
            stack.append(True)
        elif nm == 'NEWFALSE':
            stack.append(False)
        elif nm == 'TUPLE1':
            a = stack.pop(); stack.append((a,))
        elif nm == 'TUPLE2':
            b = stack.pop(); a = stack.pop(); stack.append((a, b))
        elif nm == 'TUPLE3':
            c = stack.pop(); b = stack.pop(); a = stack.pop(); stack.append((a, b, c))
        elif nm == 'TUPLE':
            stack.append(tuple(pop_to_mark()))
        elif nm == 'LIST':
            stack.append(list(pop_to_mark()))
        elif nm == 'APPEND':
            v = stack.pop(); stack[-1].append(v)
        elif nm == 'APPENDS':
            items = pop_to_mark(); stack[-1].extend(items)
        elif nm == 'SETITEM':
            v = stack.pop(); k = stack.pop(); stack[-1][k] = v
        elif nm == 'SETITEMS':
            items = pop_to_mark()
            d = stack[-1]
            for i in range(0, len(items), 2):
                d[items[i]] = items[i + 1]
        elif nm == 'STACK_GLOBAL':
            name = stack.pop(); module = stack.pop()
            stack.append(_Sym(module, name))
        elif nm == 'GLOBAL':
            module, name = arg.split(' ')
            stack.append(_Sym(module, name))
        elif nm == 'REDUCE':
            args = stack.pop(); fn = stack.pop()
            if not isinstance(fn, _Sym):
                raise ValueError(f"REDUCE on non-symbol {fn!r}")
            key = (fn.module.replace('numpy.core', 'numpy._core'), fn.name)
            if key == ('numpy._core.multiarray', '_reconstruct'):
                stack.append(_PendingArray())
            elif key == ('numpy', 'dtype'):
                stack.append(_Dtype(args[0]))
            elif key == ('numpy._core.multiarray', 'scalar'):
                dt, raw = args
                stack.append(np.frombuffer(raw, dtype=dt.np(), count=1)[0].item())
            else:
                raise ValueError(f"refusing to interpret global {fn!r}")
        elif nm == 'BUILD':
            state = stack.pop(); obj = stack[-1]
            if isinstance(obj, _Dtype):
                # (version, byteorder, subdescr, names, fields, elsize, alignment, flags)
                obj.byteorder = state[1] if state[1] in '<>' else '='
            elif isinstance(obj, _PendingArray):
                _ver, shape, dt, fortran, raw = state
                if not isinstance(raw, (bytes, bytearray)):
                    raise ValueError("object arrays are not data; refusing")
                arr = np.frombuffer(raw, dtype=dt.np()).reshape(shape, order='F' if fortran else 'C').copy()
                stack[-1] = arr
                # memo entries pointing at the placeholder must follow
                for k, v in memo.items():
                    if v is obj:
                        memo[k] = arr
            else:
                raise ValueError(f"BUILD on unexpected object {type(obj)}")
        else:
            raise ValueError(f"unsupported pickle opcode {nm}")
    raise ValueError("no STOP opcode")
