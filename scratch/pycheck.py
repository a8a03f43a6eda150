"""Names read but never bound in a Python source (a NameError waiting for the GPU box): python scratch/pycheck.py file.py ..."""
import ast, builtins, sys
for path in sys.argv[1:]:
    t = ast.parse(open(path).read())
    defined = set(dir(builtins))
    for n in ast.walk(t):
        if isinstance(n, (ast.FunctionDef, ast.ClassDef)): defined.add(n.name)
        elif isinstance(n, ast.Import):
            for a in n.names: defined.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ImportFrom):
            for a in n.names: defined.add(a.asname or a.name)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)): defined.add(n.id)
        elif isinstance(n, ast.arg): defined.add(n.arg)
        elif isinstance(n, ast.ExceptHandler) and n.name: defined.add(n.name)
    und = sorted({n.id for n in ast.walk(t) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in defined})
    print(path, "undefined:", und)
