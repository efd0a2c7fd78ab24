"""The kernel layouts every GPU parity test runs under, and the proof that the layout a test asked for is the one
that ran.

A context's step can be made of different kernels with the same results (include/mmw.h: mmw_step_kind,
mmw_kalman_layout, mmw_side_workers):

  per_scene               the bulk kernels, Kalman kernels laid out per scene (kalman_dense_min_units = -1)
  track_wise              the bulk kernels, Kalman kernels laid out over the tracks of the context
  track_wise+side_stream  ... plus the small-cloud DBSCAN workers on a second stream beside k_track
  one_workgroup           k_scene: a scene's whole track() in one workgroup

A configuration can forbid a layout (seek_inner, track_cap > 63, resized rings ...): the library then runs another
one, silently and correctly.  A test parametrised over the layouts would then be a second run of the same kernels, so
`make_checked` asks the library what it will run and SKIPS the parametrisation with the reason instead; what the
side-stream probe of the first step decided is recorded as well.  The tallies go to gpurun_out/layout_report.json
(tests/conftest.py) so that a round's test log says which kernels its green ticks stand for.
"""
import pytest

LAYOUTS = ["per_scene", "track_wise", "track_wise+side_stream", "one_workgroup"]

REPORT = {
    "contexts": {k: 0 for k in LAYOUTS},          # contexts that ran under the layout they asked for
    "skipped": [],                                 # (layout, reason) of parametrisations that would have been duplicates
    "side_probe": {"workers_on": 0, "workers_off": 0},   # what the first step's stream probe decided
    "step_kind": {},                               # mmw_step_kind() histogram over all checked contexts
}


def layout_kwargs(layout: str) -> dict:
    return {
        "kalman_dense_min_units": {"per_scene": -1, "one_workgroup": 0}.get(layout, 1),
        "chain_side_stream": 1 if layout.endswith("side_stream") else (0 if layout == "one_workgroup" else -1),
        "fused_step": 1 if layout == "one_workgroup" else -1,
    }


def make_checked(n_scenes: int, max_pts: int, layout: str, **kw):
    """SceneBatch whose kernels are those of `layout` -- or pytest.skip with the reason.  Keyword arguments override the
    layout's own (a test that sets chain_side_stream itself forbids the one-workgroup step, for instance)."""
    from mmwave_msc_amd import _lib
    from mmwave_msc_amd.batch import SceneBatch

    want = layout_kwargs(layout)
    for k, v in want.items():
        kw.setdefault(k, v)
    explicit = {k: kw[k] for k in want if kw[k] != want[k]}

    class Checked(SceneBatch):
        _first_step = True

        def _after_first_step(self):
            if not self._first_step:
                return
            self._first_step = False
            sw = self.side_workers()
            assert sw != 2, "the first mmw_step must have probed the side stream"
            if kw["chain_side_stream"] > 0 and not self.cfg.seek_inner:
                REPORT["side_probe"]["workers_on" if sw == 1 else "workers_off"] += 1

        def step_host(self, *a, **k):
            try:
                return super().step_host(*a, **k)
            finally:
                self._after_first_step()

        def step_dev(self, *a, **k):
            r = super().step_dev(*a, **k)
            self._after_first_step()
            return r

    sb = Checked(_lib.default_config(**kw), n_scenes, max_pts)
    kind, dense = sb.step_kind(), sb.kalman_layout()
    REPORT["step_kind"][str(kind)] = REPORT["step_kind"].get(str(kind), 0) + 1

    def refuse(reason):
        sb.close()
        REPORT["skipped"].append((layout, reason))
        pytest.skip(f"{layout}: {reason} -- this parametrisation would repeat another one's kernels")

    why = []
    if sb.cfg.seek_inner:
        why.append("seek_inner")
    if sb.track_cap > 63:
        why.append(f"track_cap {sb.track_cap} > 63")
    if explicit:
        why.append("the test sets " + ", ".join(f"{k}={v}" for k, v in explicit.items()))
    if layout == "one_workgroup":
        if kind != 1:
            refuse("the one-workgroup step is not available (" + ("; ".join(why) or "LDS demand") + ")")
    elif layout == "per_scene":
        assert kind in (2, 4) and dense == 0, (kind, dense)
    else:
        if dense != 1:
            refuse("the track-wise Kalman layout is not available (" + ("; ".join(why) or "?") + ")")
        assert kind == 4, kind
        if layout.endswith("side_stream") and "chain_side_stream" not in explicit and sb.side_workers() == 0:
            refuse("the side-stream workers are not available (" + ("; ".join(why) or "?") + ")")
    REPORT["contexts"][layout] += 1
    return sb
