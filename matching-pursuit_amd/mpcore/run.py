"""Run a script of the reference under the overlay, without editing it:

    cd /path/to/matching-pursuit            # the reference checkout
    PYTHONPATH=/path/to/matching-pursuit_amd python -m mpcore.run iterativedecomposition.py [script args]
    PYTHONPATH=/path/to/matching-pursuit_amd python -m mpcore.run --multiband mp.py

Equivalent to `python script.py` after `import mpcore; mpcore.install()` (mpcore/overlay.py): the script's
directory goes to the front of sys.path as the interpreter would put it, so `import modules` finds the
reference's package there; the hot-path names in it are mpcore's, everything else is the reference's.
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    multiband = False
    while argv and argv[0].startswith("--"):
        flag = argv.pop(0)
        if flag == "--multiband":
            multiband = True
        elif flag == "--":
            break
        else:
            raise SystemExit(f"mpcore.run: unknown option {flag}\n{__doc__}")
    if not argv:
        raise SystemExit(__doc__)
    script = argv[0]
    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    from . import overlay
    mode = overlay.install(multiband=multiband)
    print(f"[mpcore] {mode} installed over `modules`", file=sys.stderr)
    sys.argv = argv
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
