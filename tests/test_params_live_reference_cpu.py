"""The command line against the reference's own parser, live (build container only: skipped without /root/reference).

`consistent_depth_amd.params.Video3dParamsParser` promises "same flags, defaults and post-parse resolution" as
/root/reference/params.py:15-123.  Here the reference's parser itself is imported -- with oracle/ref_loop's stub set plus a stand-in
for the absent COLMAP submodule and a path-only `tools` package (this repo's tools/ would shadow the reference's) -- and both
parsers parse the same command lines: every flag of the reference exists here with the same default, and the resolved
namespaces agree on every key the reference produces (model-dependent defaults, --configure kitti, frame ranges)."""
import importlib.abc
import importlib.machinery
import os
import sys
import types

import pytest

from oracle import ref_loop

pytestmark = pytest.mark.skipif(not ref_loop.available(), reason="the reference checkout is not on this machine")

CASES = [
    ["--path", "/tmp/x"],
    ["--path", "/tmp/x", "--model_type", "midas2"],
    ["--path", "/tmp/x", "--configure", "kitti"],
    ["--path", "/tmp/x", "--learning_rate", "1e-5", "--lambda_view_baseline", "0.3", "--align", "32", "--frame_range", "0-3,7",
     "--flow_ops", "hierarchical2", "consecutive", "--batch_size", "2", "--num_epochs", "3", "--size", "224", "--make_video"],
    ["--path", "/tmp/x", "--op", "extract_frames", "--video_file", "v.mp4", "--sparse", "--matcher", "sequential", "--lambda_parameter", "0.5",
     "--optimizer", "Adam", "--val_epoch_freq", "2", "--save_epoch_freq", "5", "--overlap_ratio", "0.4"],
]


class _ThirdPartyStub(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """third_party/colmap is an un-vendored submodule: any `third_party.*` import gets a module of inert names."""

    def find_spec(self, name, path, target=None):
        if name == "third_party" or name.startswith("third_party."):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        m.__getattr__ = lambda attr: [] if attr.isupper() else type(attr, (), {})
        return m

    def exec_module(self, module):
        pass


def _plain(v):
    if hasattr(v, "name") and hasattr(v, "set"):            # NamedOptionalSet of either code base
        return ("range", v.name)
    return v


def _reference_namespaces():
    import torch
    top = ref_loop._REF_TOP
    ref_loop._REF_TOP = tuple(top) + ("tools", "params", "scale_calibration", "flow", "video", "process", "third_party")
    stub = _ThirdPartyStub()
    argv = sys.argv
    try:
        with ref_loop.reference_modules(torch.float32):
            sys.meta_path.insert(0, stub)
            tools = types.ModuleType("tools")
            tools.__path__ = [os.path.join(ref_loop.REF, "tools")]
            sys.modules["tools"] = tools
            import params as ref_params
            opts, out = None, []
            for case in CASES:
                parser = ref_params.Video3dParamsParser()
                sys.argv = ["main.py"] + case
                ns = parser.parse()
                if opts is None:
                    opts = {a.option_strings[0]: a.default for a in parser.parser._actions if a.option_strings and a.option_strings[0] != "-h"}
                out.append({k: _plain(v) for k, v in vars(ns).items()})
            return opts, out
    finally:
        sys.argv = argv
        if stub in sys.meta_path:
            sys.meta_path.remove(stub)
        ref_loop._REF_TOP = top


def test_flags_defaults_and_resolution_match_the_reference_parser(capsys):
    from consistent_depth_amd.params import Video3dParamsParser
    ref_opts, ref_ns = _reference_namespaces()
    ours = Video3dParamsParser()
    ours.initialize()
    our_opts = {a.option_strings[0]: a.default for a in ours.parser._actions if a.option_strings and a.option_strings[0] != "-h"}
    missing = sorted(set(ref_opts) - set(our_opts))
    assert not missing, f"flags of the reference's command line that this parser does not take: {missing}"
    for flag, default in ref_opts.items():
        assert _plain(our_opts[flag]) == _plain(default), (flag, our_opts[flag], default)
    assert sorted(set(our_opts) - set(ref_opts)) == ["--seed"]        # the one extension (seed of the permutations / random init)
    for case, ref in zip(CASES, ref_ns):
        got = {k: _plain(v) for k, v in vars(Video3dParamsParser().parse(case)).items()}
        for key, value in ref.items():
            assert key in got and got[key] == value, (case, key, got.get(key), value)
    capsys.readouterr()
