#!/usr/bin/env python3
"""Prototype-only C declarations of include/monorun_pnp.h in the form `cffi.FFI().cdef()` accepts (no preprocessor lines other
than plain integer #defines, no comments, no extern "C").  INTEGRATION.md §3 carries the output of this script between its
`cffi-cdef` markers; tests/test_capi_and_host.py checks that the two agree and that every prototype is an exported symbol.

    python tools/gen_cffi_cdef.py            # prints the cdef text
    python tools/gen_cffi_cdef.py --update   # rewrites the block in INTEGRATION.md
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_prototypes(path=os.path.join(ROOT, 'include', 'monorun_pnp.h')):
    """[(name, 'return type', ['arg type name', ...])] for every function the header declares."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    src = re.sub(r'//[^\n]*', ' ', src)
    src = '\n'.join(l for l in src.split('\n') if not l.lstrip().startswith('#') and 'extern "C"' not in l and l.strip() != '}')
    out = []
    for m in re.finditer(r'([A-Za-z_][\w\s\*]*?)\b(\w+)\s*\(([^()]*)\)\s*;', src):
        ret, name, args = ' '.join(m.group(1).split()), m.group(2), m.group(3)
        args = [' '.join(a.split()).replace(' *', ' *').replace('* ', '*') for a in args.split(',')] if args.strip() not in ('', 'void') else []
        out.append((name, ret, args))
    return out


def header_int_defines(path=os.path.join(ROOT, 'include', 'monorun_pnp.h')):
    """plain integer #defines (cffi accepts `#define NAME <integer literal>` only: negative values and shifted masks are
    left out and listed as Python constants in monorun_amd/_lib.py instead)"""
    out = []
    for l in open(path):
        m = re.match(r'#define\s+(MR_\w+)\s+(0x[0-9A-Fa-f]+|\d+)\s*(/\*.*)?$', l.strip())
        if m:
            out.append((m.group(1), m.group(2)))
    return out


def cdef_text():
    lines = [f'#define {n} {v}' for n, v in header_int_defines()]
    for name, ret, args in header_prototypes():
        lines.append(f'{ret} {name}({", ".join(args) if args else "void"});')
    return '\n'.join(lines) + '\n'


def update_integration_md(path=os.path.join(ROOT, 'INTEGRATION.md')):
    """Rewrite the text between the `cffi-cdef` markers of INTEGRATION.md with the generated prototypes."""
    src = open(path).read()
    a, b = src.index('<!-- cffi-cdef:begin -->'), src.index('<!-- cffi-cdef:end -->')
    head = src[:a] + '<!-- cffi-cdef:begin -->\n```c\n'
    open(path, 'w').write(head + cdef_text() + '```\n' + src[b:])


if __name__ == '__main__':
    import sys
    if '--update' in sys.argv:
        update_integration_md()
    else:
        print(cdef_text(), end='')
