"""Writes tests/golden/reference_api_names.json: the NAMES (modules, classes, methods, functions and their positional parameter names -- no code) of the
reference package's public surface, read from its sources with `ast` in the build container.  The CPU suite checks
that every name resolves in this package (tests/test_host_logic.py::test_every_reference_name_resolves)."""
import ast
import json
import os

REF = '/root/reference/LinearResponseVariationalBayes'
out = {}
for f in sorted(os.listdir(REF)):
    if not f.endswith('.py') or f.startswith('test') or f == '__init__.py':
        continue
    tree = ast.parse(open(os.path.join(REF, f)).read())
    def args_of(fn):
        return [a.arg for a in fn.args.args]              # positional parameter NAMES, in order

    names = {'functions': {}, 'classes': {}}
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            names['classes'][node.name] = {n.name: args_of(n) for n in node.body if isinstance(n, ast.FunctionDef)
                                           and (not n.name.startswith('__') or n.name == '__init__')}
        elif isinstance(node, ast.FunctionDef):
            names['functions'][node.name] = args_of(node)
    out[f[:-3]] = names
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_api_names.json'), 'w') as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print({k: (len(v['functions']), len(v['classes'])) for k, v in out.items()})
