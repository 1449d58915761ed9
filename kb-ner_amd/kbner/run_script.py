"""Run an UNCHANGED entry script of the reference checkout (train.py, …) against this package's `flair`.

    PYTHONPATH=/path/to/repo/kb-ner_amd python -m kbner.run_script /path/to/KB-NER/train.py --config config/<your>.yaml

Why not `python /path/to/KB-NER/train.py`: the interpreter puts the SCRIPT's directory in sys.path[0], ahead of PYTHONPATH, and
the reference's own `flair/` lives there -- its `import flair` would pick the reference package (and fail on the first module the
image lacks, e.g. segtok in flair/data.py).  `runpy.run_path` executes the file as `__main__` without touching sys.path, so the
`flair` found first on PYTHONPATH -- this mirror -- is the one the script imports.  Relative paths inside the script (config
files, `resources/…`) resolve against the current working directory exactly as they do for the reference."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        sys.stderr.write(__doc__ + "\n")
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        sys.stderr.write("kbner.run_script: no such script: %s\n" % script)
        return 2
    # the script's own directory must not shadow the mirror (sys.path[0] of `python -m` is the cwd / '' entry: drop it when it
    # is the reference checkout itself)
    sdir = os.path.dirname(script)
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != sdir or
                   os.path.isfile(os.path.join(os.path.abspath(p or os.getcwd()), "kbner", "__init__.py"))]
    import flair  # noqa: F401  (fail here, loudly, if the mirror is not importable)
    if os.path.abspath(os.path.dirname(os.path.dirname(flair.__file__))) == sdir:
        sys.stderr.write("kbner.run_script: `flair` resolved to the script's own directory (%s); put kb-ner_amd on PYTHONPATH\n" % sdir)
        return 2
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
