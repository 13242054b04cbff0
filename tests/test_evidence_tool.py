"""tools/evidence.py -- the hygiene rule of profiles/roundN: a log or counter file is only published when it says that it was measured on
the kernel sources the tree holds (evidence older than the shipped kernel was labelled as current)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("evidence", os.path.join(ROOT, "tools", "evidence.py"))
evidence = importlib.util.module_from_spec(spec)
spec.loader.exec_module(evidence)


def test_sha_is_the_hash_bench_compares_with():
    import bench
    assert evidence.sha() == bench.kernel_source_sha() and len(evidence.sha()) == 16


def test_publish_takes_current_files_and_refuses_stale_ones(tmp_path, capsys):
    cur = evidence.sha()
    good_log = tmp_path / "a.log"; good_log.write_text("rows\n"); evidence.stamp(str(good_log))
    assert good_log.read_text().startswith(f"# kernel_source_sha={cur} ")
    stale_log = tmp_path / "b.log"; stale_log.write_text("# kernel_source_sha=0123456789abcdef 2026-01-01 00:00:00 b.log\nrows\n")
    evidence.stamp(str(stale_log))                                    # a stamped log is never re-stamped after the fact
    assert stale_log.read_text().startswith("# kernel_source_sha=0123456789abcdef")
    bare_log = tmp_path / "c.log"; bare_log.write_text("no header\n")
    good_json = tmp_path / "sq.json"; good_json.write_text(json.dumps({"kernel_source_sha": cur, "x": 1}))
    stale_json = tmp_path / "tr.json"; stale_json.write_text(json.dumps({"kernel_source_sha": "0123456789abcdef"}))
    csv = tmp_path / "k.csv"; csv.write_text("Name,Calls\n"); (tmp_path / "k.csv.sha").write_text(cur + "\n")
    bench_line = tmp_path / "bench.json"; bench_line.write_text(json.dumps({"metric": "m", "value": 1.0}) + "\n"); (tmp_path / "bench.json.sha").write_text(cur + "\n")
    dst = tmp_path / "round"
    rc = evidence.publish(str(dst), [str(p) for p in (good_log, stale_log, bare_log, good_json, stale_json, csv, bench_line)])
    assert rc == 1
    assert sorted(os.listdir(dst)) == ["a.log", "bench.json", "bench.json.sha", "k.csv", "k.csv.sha", "sq.json"]
    err = capsys.readouterr().err
    assert "REFUSED" in err and "b.log" in err and "c.log" in err and "tr.json" in err
    assert evidence.publish(str(dst), [str(good_log)]) == 0


def test_everything_under_profiles_round4_names_its_sources():
    """Every judged file of the round carries the hash of the sources it was measured on (header, key or sidecar)."""
    d = os.path.join(ROOT, "profiles", "round4")
    missing = []
    for name in sorted(os.listdir(d)):
        if name.endswith((".sha", ".md")):
            continue
        if evidence.stamp_of(os.path.join(d, name)) is None:
            missing.append(name)
    assert missing == []


def test_code_object_metadata_of_the_step_kernels():
    lib = os.path.join(ROOT, "ft_grandprix_amd", "lib", "libftgp.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(evidence.LLVM, "llvm-readelf")):
        import pytest
        pytest.skip("library or llvm tools not present")
    meta = evidence.code_object_meta(lib)
    assert len(meta) == 6                                          # <MULTI, FAKE, ROSTER>: the six instantiations
    for name, m in meta.items():
        assert m["vgpr_count"] <= 64, name                        # 8 waves per SIMD
        assert m["scratch_bytes_per_lane"] == 0 and m["vgpr_spill_count"] == 0, name
