"""Host-side mirror of Crucible's keyframe authoring API, flattened to CrKeyframe.

Mirrors `TransformTimeline` (reference src/timeline/mod.rs:116-231) and its
builders `translate_{x,y,z,point}` / `scale_sphere`
(src/timeline/transform_builder.rs).  The reference stores 4x4 matrices of
closures; only three closure shapes ever appear on the translate / radius
channels (`move |_t| c`, `move |t| x * t`, `move |t| s + (r - s) * t`), so each
Transform flattens to one CrKeyframe (include/crucible_hip.h).  The per-axis
scale builders `scale_{x,y,z,point}` (transform_builder.rs:101-346,729) flatten to
the CR_KEY_SCALE_* channels; which matrix slot each writes (ScaleY: row 1,
column 0) is applied where the keys are evaluated (pathtrace.hpp scale_point).
"""
from dataclasses import dataclass

from . import _abi as A

NERP = "NERP"   # InterpolationType::NERP, timeline/mod.rs:100-103
LERP = "LERP"
WORLD = "World"  # TransformSpace, timeline/mod.rs:108-111
LOCAL = "Local"

_OMNI = "Omni"


@dataclass
class _Transform:
    channel: int       # CR_KEY_TX.. or CR_KEY_RADIUS; -1 for the initial Omni transform
    ttype: str         # "TranslateX" | ... | "ScaleR" | "Omni"
    t0: float
    t1: float
    interp: int
    a: float
    b: float
    end: object        # TransformResult: ("TranslateX", v) | ("InitTranslate", (x,y,z)) | ("ScaleR", v) | ("InitScale", v)


class TransformTimeline:
    """TransformTimeline::new / new_sphere (timeline/mod.rs:127-231)."""

    def __init__(self, start_pos, start_scale=1.0, sphere=False):
        self.start_pos = tuple(float(c) for c in start_pos)
        self.start_scale = float(start_scale)
        self.sphere = sphere
        self.scale = [_Transform(-1, _OMNI, -0.1, -0.1, A.CR_KEY_NERP, self.start_scale, 0.0,
                                 ("InitScale", self.start_scale))]
        self.translate = [_Transform(-1, _OMNI, -0.1, -0.1, A.CR_KEY_NERP, 0.0, 0.0,
                                     ("InitTranslate", self.start_pos))]

    @classmethod
    def new_sphere(cls, start_pos, start_radius):
        return cls(start_pos, start_radius, sphere=True)

    def clone(self):
        import copy
        return copy.deepcopy(self)

    # helper_functions.rs:41-140
    @staticmethod
    def _most_recent(lst, t, ttype):
        for tf in reversed(lst):
            if t > tf.t1 and tf.ttype in (ttype, _OMNI):   # valid_time.is_less(t)
                return tf
        return None

    def _translate_axis(self, axis, x, keyframe, interp, space):
        # translate_x/y/z, transform_builder.rs:339-717
        assert keyframe >= 0.0, "Cannot add a keyframe before the animation start."
        ttype = ("TranslateX", "TranslateY", "TranslateZ")[axis]
        prev = self._most_recent(self.translate, keyframe, ttype)
        if prev is None:
            raise ValueError("Missing transform data! could not find a previous position reference")
        prev_end = prev.end
        prev_time = max(prev.t1, 0.0)
        standard = x
        if space == WORLD:
            start = prev_end[1] if prev_end[0] == ttype else prev_end[1][axis]
            x = x - start
        if interp == LERP:
            tf = _Transform(axis, ttype, prev_time, keyframe, A.CR_KEY_LERP, x, 0.0, (ttype, standard))
        else:
            tf = _Transform(axis, ttype, keyframe, keyframe, A.CR_KEY_NERP, x, 0.0, (ttype, standard))
        self.translate.append(tf)
        self.translate.sort(key=lambda t: t.t0)   # stable, compare_start (utils.rs:680-686)

    def translate_x(self, x, keyframe, interp, space):
        self._translate_axis(0, float(x), float(keyframe), interp, space)

    def translate_y(self, y, keyframe, interp, space):
        self._translate_axis(1, float(y), float(keyframe), interp, space)

    def translate_z(self, z, keyframe, interp, space):
        self._translate_axis(2, float(z), float(keyframe), interp, space)

    def translate_point(self, p, keyframe, interp, space):   # transform_builder.rs:721-731
        self.translate_x(p[0], keyframe, interp, space)
        self.translate_y(p[1], keyframe, interp, space)
        self.translate_z(p[2], keyframe, interp, space)

    def scale_sphere(self, r, keyframe, interp):   # transform_builder.rs:17-96
        assert keyframe >= 0.0, "Cannot add a keyframe before the animation start."
        r = float(r)
        keyframe = float(keyframe)
        prev = self._most_recent(self.scale, keyframe, "ScaleR")
        if prev is None:
            raise ValueError("Missing transform data! Tried to scale radius but could not find a previous scale reference!")
        prev_end = prev.end
        prev_time = max(prev.t1, 0.0)
        if interp == LERP:
            start = prev_end[1]
            tf = _Transform(A.CR_KEY_RADIUS, "ScaleR", prev_time, keyframe, A.CR_KEY_LERP, start, r, ("ScaleR", r))
        else:
            tf = _Transform(A.CR_KEY_RADIUS, "ScaleR", keyframe, keyframe, A.CR_KEY_NERP, r, 0.0, ("ScaleR", r))
        self.scale.append(tf)
        self.scale.sort(key=lambda t: t.t0)

    def _scale_axis(self, axis, x, keyframe, interp):
        # scale_x / scale_y / scale_z, transform_builder.rs:101-346: the previous ScaleX|Y|Z (or the initial Omni)
        # transform gives the start value and, for LERP, the start time
        assert keyframe >= 0.0, "Cannot add a keyframe before the animation start."
        ttype = ("ScaleX", "ScaleY", "ScaleZ")[axis]
        channel = (A.CR_KEY_SCALE_X, A.CR_KEY_SCALE_Y, A.CR_KEY_SCALE_Z)[axis]
        x, keyframe = float(x), float(keyframe)
        prev = self._most_recent(self.scale, keyframe, ttype)
        if prev is None:
            raise ValueError(f"Missing transform data! Tried to scale {'xyz'[axis]} but could not find a previous scale reference!")
        prev_end = prev.end
        prev_time = max(prev.t1, 0.0)
        if interp == LERP:
            start = prev_end[1]   # ScaleX(start) | InitScale(start)
            tf = _Transform(channel, ttype, prev_time, keyframe, A.CR_KEY_LERP, start, x, (ttype, x))
        else:
            tf = _Transform(channel, ttype, keyframe, keyframe, A.CR_KEY_NERP, x, 0.0, (ttype, x))
        self.scale.append(tf)
        self.scale.sort(key=lambda t: t.t0)

    def scale_x(self, x, keyframe, interp):
        self._scale_axis(0, x, keyframe, interp)

    def scale_y(self, y, keyframe, interp):
        self._scale_axis(1, y, keyframe, interp)

    def scale_z(self, z, keyframe, interp):
        self._scale_axis(2, z, keyframe, interp)

    def scale_point(self, p, keyframe, interp):   # transform_builder.rs:729-733
        self.scale_x(p[0], keyframe, interp)
        self.scale_y(p[1], keyframe, interp)
        self.scale_z(p[2], keyframe, interp)

    def is_static(self):
        return len(self.scale) == 1 and len(self.translate) == 1

    def keyframes(self):
        """Flatten to CrKeyframe records: translate list order, then scale list order."""
        out = []
        for tf in self.translate[0:]:
            if tf.channel >= 0:
                out.append(A.CrKeyframe(tf.channel, tf.interp, tf.t0, tf.t1, tf.a, tf.b))
        for tf in self.scale:
            if tf.channel >= 0:
                out.append(A.CrKeyframe(tf.channel, tf.interp, tf.t0, tf.t1, tf.a, tf.b))
        return out
