# poor man's pyflakes: names loaded but never bound anywhere in the module / builtins
import ast, builtins, sys
for path in sys.argv[1:]:
    t = ast.parse(open(path).read())
    bound = set(dir(builtins))
    for n in ast.walk(t):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names: bound.add((a.asname or a.name).split('.')[0])
        elif isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            bound.add(n.name)
            if not isinstance(n, ast.ClassDef):
                for a in n.args.args + n.args.kwonlyargs + n.args.posonlyargs: bound.add(a.arg)
                if n.args.vararg: bound.add(n.args.vararg.arg)
                if n.args.kwarg: bound.add(n.args.kwarg.arg)
        elif isinstance(n, ast.Lambda):
            for a in n.args.args + n.args.kwonlyargs: bound.add(a.arg)
            if n.args.vararg: bound.add(n.args.vararg.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
        elif isinstance(n, ast.comprehension):
            for m in ast.walk(n.target):
                if isinstance(m, ast.Name): bound.add(m.id)
    for n in ast.walk(t):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound:
            print(f"{path}:{n.lineno}: undefined name {n.id}")
