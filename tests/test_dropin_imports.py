"""CPU tests of the drop-in module paths (SURVEY.md section 8 b): the `lib/` package here serves the hot-path modules
and lets every module it does not serve fall through to a reference checkout later on sys.path
(run_inference_torch_data.py:13-31 imports lib.data_utils.{async_dataset,dataset_util,split} next to the served ones).

The fall-through itself is tested with a synthetic sibling tree; the run against the real reference scripts only
happens where /root/reference exists (the build container) and is skipped elsewhere."""
import os
import shutil
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"


def _run(code, pythonpath, cwd):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(pythonpath))
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, cwd=cwd, capture_output=True,
                          text=True, timeout=300)


def test_unserved_modules_fall_through_to_a_sibling_lib_tree(tmp_path):
    # a "reference checkout": namespace packages (no __init__.py), one module we do not serve at each level and one
    # we do serve (must lose against this repo's)
    sib = tmp_path / "checkout"
    (sib / "lib" / "data_utils").mkdir(parents=True)
    (sib / "lib" / "extra_pkg").mkdir(parents=True)
    (sib / "lib" / "tracker").mkdir(parents=True)
    (sib / "lib" / "data_utils" / "extra_mod.py").write_text(
        "from lib.data_utils import fs\nfrom .idxbinfile import TorchIdx\nWHO = 'sibling'\n")
    (sib / "lib" / "extra_pkg" / "mod.py").write_text("WHO = 'sibling pkg'\n")
    (sib / "lib" / "data_utils" / "fs.py").write_text("WHO = 'shadowed'\n")
    (sib / "lib" / "tracker" / "tracker.py").write_text("WHO = 'shadowed'\n")
    r = _run("""
        import lib.data_utils.extra_mod as e, lib.extra_pkg.mod as m
        import lib.data_utils.fs as fs, lib.tracker.tracker as t, lib.data_utils.idxbinfile as ib
        import absolutetrack_amd.formats as f
        assert e.WHO == 'sibling' and m.WHO == 'sibling pkg'
        assert e.fs is fs and not hasattr(fs, 'WHO') and hasattr(fs, 'read_bytes')
        assert e.TorchIdx is f.TorchIdx is ib.TorchIdx
        assert hasattr(t, 'HandTracker') and not hasattr(t, 'WHO')
        print(e.__file__, '|', t.__file__)
        """, [ROOT, str(sib)], str(tmp_path))
    assert r.returncode == 0, r.stderr
    ext, trk = r.stdout.strip().split(" | ")
    assert ext.startswith(str(sib)) and trk.startswith(ROOT)
    # without the sibling on the path the unserved module is simply absent
    r = _run("import lib.data_utils.extra_mod", [ROOT], str(tmp_path))
    assert r.returncode != 0 and "ModuleNotFoundError" in r.stderr


needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "lib")),
                                     reason="reference checkout not present (GPU box)")


@needs_reference
@pytest.mark.parametrize("script", ["run_inference_torch_data.py", "load_eval.py"])
def test_reference_scripts_import_on_the_dropin(script, tmp_path):
    """`PYTHONPATH=<repo>:<reference> python <reference script>` gets past its imports (run under another
    __name__, so the __main__ block -- which needs the absent dataset / weights -- does not run).
    run_eval_*_skeleton.py also import PyAV (`import av`), which is not installed here."""
    r = _run(f"""
        import runpy
        ns = runpy.run_path({os.path.join(REFERENCE, script)!r}, run_name='imported')
        import lib.models.model_loader as ml, lib.data_utils.bundles as b
        assert ml.__file__.startswith({ROOT!r}) and b.__file__.startswith({ROOT!r})
        if 'find_dataset' in ns:
            import lib.data_utils.async_dataset as a
            assert a.__file__.startswith({REFERENCE!r})
            import absolutetrack_amd.torch_data as td, absolutetrack_amd.model as m
            assert ns['preprocess'] is td.preprocess and ns['load_pretrained_model'] is m.load_pretrained_model
        print('ok')
        """, [ROOT, REFERENCE], str(tmp_path))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr


@needs_reference
def test_reference_dataset_loader_reads_through_the_served_idxbinfile(tmp_path, golden_dir):
    """The reference's own loader chain (find_dataset -> AsyncToIterableDataset, run_inference_torch_data.py:143-171)
    runs on top of the TorchIdx / fs served here and yields the same raw sequences as formats.read_sequence.
    The reference's Sampler passes an argument to torch.utils.data.Sampler.__init__, which this image's torch 2.10 no
    longer accepts (its own incompatibility, nothing served here is involved), so a plain index range stands in."""
    leaf = tmp_path / "torch_data" / "real" / "testing"
    leaf.mkdir(parents=True)
    for field, src in (("mono", "seq_mono"), ("labels", "seq_labels")):
        for ext in (".torch.idx", ".torch.bin"):
            shutil.copy(os.path.join(golden_dir, src + ext), leaf / (field + ext))
    out = tmp_path / "got.npz"
    r = _run(f"""
        import numpy as np, msgpack
        from lib.data_utils.async_dataset import AsyncToIterableDataset, find_dataset
        from lib.data_utils.split import Split
        ds = find_dataset([{str(tmp_path / "torch_data" / "real")!r}], ["mono", "labels"])
        d = ds[Split.TEST]
        it = AsyncToIterableDataset(d, range(len(d)), max_prefetch=4)
        rows = list(it)
        np.savez({str(out)!r}, n=len(rows), **{{f"mono{{i}}": np.asarray(r["mono"]) for i, r in enumerate(rows)}},
                 **{{f"lab{{i}}": np.frombuffer(msgpack.packb(r["labels"]), np.uint8) for i, r in enumerate(rows)}})
        """, [ROOT, REFERENCE], str(tmp_path))
    assert r.returncode == 0, r.stderr
    import msgpack
    from absolutetrack_amd import formats
    got = np.load(out)
    assert int(got["n"]) == 3
    for i in range(3):
        want = formats.read_sequence(str(leaf / "mono.torch.idx"), str(leaf / "labels.torch.idx"), i)
        assert np.array_equal(got[f"mono{i}"], want["mono"])
        assert msgpack.unpackb(got[f"lab{i}"].tobytes()) == want["labels"]
