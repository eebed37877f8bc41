"""The build is proven by content, not by mtimes (latent-flexible-video-diffusion-modeling_amd/build.py): an untouched tree reuses every
object, a changed header digest recompiles every object, a tampered object file is never linked."""
import os
import shutil

import __graft_entry__ as entry


def _all_sources(mod):
    return sorted(os.path.basename(s)[:-4] for s in mod.source_files())


def test_build_reuses_by_content_and_recompiles_on_header_change(tmp_path):
    rep = entry.build()                                   # the in-tree library (what travels to the GPU box)
    mod = entry._build_module()
    names = _all_sources(mod)
    assert sorted(rep["compiled"] + rep["reused"]) == names
    rep2 = entry.build()                                  # nothing edited: everything is reused, nothing is linked
    assert sorted(rep2["reused"]) == names and rep2["compiled"] == [] and not rep2["linked"]

    # a copy of lib/ stands for "the same tree on another box": mtimes differ (copy), contents do not
    work = str(tmp_path / "lib")
    shutil.copytree(mod.LIBDIR, work)
    for f in os.listdir(work):
        os.utime(os.path.join(work, f), (1, 1))           # ancient mtimes must not matter either way
    mod.build(verbose=False, libdir=work)
    rep3 = mod.build_report()
    assert sorted(rep3["reused"]) == names and rep3["compiled"] == []

    # a tampered object is detected by its digest and recompiled (only that one), and the library is relinked
    victim = os.path.join(work, names[-1] + ".o")
    with open(victim, "ab") as f:
        f.write(b"\0")
    mod.build(verbose=False, libdir=work)
    rep4 = mod.build_report()
    assert rep4["compiled"] == [names[-1]] and rep4["linked"]

    # a header whose content hash changed: every object is recompiled
    mod.build(verbose=False, libdir=work, header_salt="changed")
    rep5 = mod.build_report()
    assert sorted(rep5["compiled"]) == names and rep5["reused"] == [] and rep5["linked"]


def test_force_build_env(monkeypatch, tmp_path):
    mod = entry._build_module()
    work = str(tmp_path / "lib")
    shutil.copytree(mod.LIBDIR, work)
    mod.build(verbose=False, libdir=work, force=True)
    assert sorted(mod.build_report()["compiled"]) == _all_sources(mod)
    # __graft_entry__.build() reads LFVDM_FORCE_BUILD
    import inspect
    assert "LFVDM_FORCE_BUILD" in inspect.getsource(entry.build)
