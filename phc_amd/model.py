"""Model compiler: MJCF -> flat articulation description consumed by the HIP stepper.

Replaces what the reference gets from Isaac Gym's asset pipeline
(``gym.load_asset / get_actor_dof_properties / get_actor_rigid_body_properties``,
reference ``phc/env/tasks/humanoid.py:768-990,1093-1106``) and from
``SkeletonTree.from_mjcf`` (``poselib/poselib/skeleton/skeleton3d.py:149-193``):

* body order = MJCF depth-first order (identical to ``SkeletonTree.from_mjcf``);
* the three co-located hinges of a body are merged into one spherical joint whose
  coordinates are the exponential map of the child-in-parent rotation
  (Isaac Gym convention the reference relies on, ``humanoid.py:487,1762``);
* mass / centre of mass / inertia come from geom density x volume
  (sphere, capsule ``fromto``, box), like MuJoCo / Isaac Gym do for MJCF without
  ``<inertial>``;  ``<inertial>`` is honoured when present (robots);
* ground-contact primitives: sphere -> 1 point (r), capsule -> 2 points (r),
  box -> 8 corners (r = 0).

The compiled model is a plain dict of numpy arrays / lists, serialisable to JSON
(``phc_amd/assets/*.json`` are compiled once from the reference's MJCF assets by
``python -m phc_amd.model <mjcf> <out.json>``; the assets are data, the reference
tree is not needed at run time).
"""
import json
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

JOINT_FREE, JOINT_SPHERICAL, JOINT_REVOLUTE, JOINT_FIXED = 0, 1, 2, 3
ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def _floats(s, default=None):
    if s is None:
        return None if default is None else np.array(default, dtype=np.float64)
    return np.array([float(x) for x in s.split()], dtype=np.float64)


def _quat_wxyz_to_mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _shift_inertia(I_c, m, c):
    """Inertia about a point displaced by -c from the COM (parallel axis)."""
    return I_c + m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))


_HULL_DIRS = np.array([d for d in np.ndindex(3, 3, 3)], dtype=np.float64) - 1.0
_HULL_DIRS = _HULL_DIRS[(np.abs(_HULL_DIRS).sum(1) == 1) | (np.abs(_HULL_DIRS).sum(1) == 3)]  # 6 axes + 8 diagonals


def _stl_vertices(path):
    """Vertices of a binary STL (80-byte header, uint32 count, 50-byte records)."""
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw, dtype="<u4", count=1, offset=80)[0])
    rec = np.frombuffer(raw, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    return rec["v"].reshape(-1, 3).astype(np.float64)


def _mesh_contact_points(verts):
    """Support points of the mesh's convex hull along 14 directions (Isaac Gym collides the convex hull of a URDF mesh):
    what ground contact and the start-height fix need of the link's shape."""
    idx = sorted({int(np.argmax(verts @ d)) for d in _HULL_DIRS})
    return [(verts[i], 0.0) for i in idx]


def _geom_props(g, defaults, meshes=None):
    """-> (mass, com[3], I_com[3,3], contact points [(pos[3], radius)])"""
    gtype = g.attrib.get("type", defaults.get("type", "sphere"))
    if gtype == "mesh":
        # robots carry <inertial>; the mesh only contributes contact points (pos/quat of the geom applied)
        verts = meshes[g.attrib["mesh"]]
        pos = _floats(g.attrib.get("pos"), [0, 0, 0])
        R = _quat_wxyz_to_mat(_floats(g.attrib.get("quat"), [1, 0, 0, 0]))
        return 0.0, pos, np.zeros((3, 3)), [(pos + R @ p, r) for p, r in _mesh_contact_points(verts)]
    if gtype == "cylinder":
        # asset_options.replace_cylinder_with_capsule = True (humanoid.py:894): two sphere points on the axis
        g = _Attr(dict(g.attrib, type="capsule"))
        gtype = "capsule"
    density = float(g.attrib.get("density", defaults.get("density", 1000.0)))
    size = _floats(g.attrib.get("size"), [0.0])
    if gtype == "sphere":
        r = size[0]
        pos = _floats(g.attrib.get("pos"), [0, 0, 0])
        m = density * 4.0 / 3.0 * np.pi * r ** 3
        I = np.eye(3) * 0.4 * m * r * r
        return m, pos, I, [(pos, r)]
    if gtype == "capsule":
        r = size[0]
        if "fromto" in g.attrib:
            ft = _floats(g.attrib["fromto"])
            p0, p1 = ft[:3], ft[3:]
        else:
            pos = _floats(g.attrib.get("pos"), [0, 0, 0])
            R = _quat_wxyz_to_mat(_floats(g.attrib.get("quat"), [1, 0, 0, 0]))
            half = size[1]
            p0, p1 = pos - R[:, 2] * half, pos + R[:, 2] * half
        L = np.linalg.norm(p1 - p0)
        axis = (p1 - p0) / max(L, 1e-12)
        m_cyl = density * np.pi * r * r * L
        m_cap = density * 4.0 / 3.0 * np.pi * r ** 3
        I_ax = 0.5 * m_cyl * r * r + 0.4 * m_cap * r * r
        I_tr = m_cyl * (L * L / 12.0 + r * r / 4.0) + m_cap * (0.4 * r * r + L * L / 4.0 + 0.375 * L * r)
        P = np.outer(axis, axis)
        I = I_ax * P + I_tr * (np.eye(3) - P)
        return m_cyl + m_cap, 0.5 * (p0 + p1), I, [(p0, r), (p1, r)]
    if gtype == "box":
        pos = _floats(g.attrib.get("pos"), [0, 0, 0])
        R = _quat_wxyz_to_mat(_floats(g.attrib.get("quat"), [1, 0, 0, 0]))
        a, b, c = size[:3]
        m = density * 8.0 * a * b * c
        I_l = np.diag([b * b + c * c, a * a + c * c, a * a + b * b]) * m / 3.0
        I = R @ I_l @ R.T
        pts = []
        for sx in (-1, 1):
            for sy in (-1, 1):
                for sz in (-1, 1):
                    pts.append((pos + R @ np.array([sx * a, sy * b, sz * c]), 0.0))
        return m, pos, I, pts
    raise NotImplementedError(f"geom type {gtype}")


class _Attr:
    def __init__(self, attrib):
        self.attrib = attrib


def _geom_capsules(g, defaults, pts):
    """Collision capsules [(a[3], b[3], radius)] standing in for a geom in body-body contact: spheres and capsules are exact; a box becomes
    the capsule along its longest axis with the mean of the two other half extents as radius -- or, when it is FLAT (the larger of those two
    half extents exceeds 1.5 x the smaller: the SMPL toes and hands), TWO capsules of the smaller half extent side by side (round 4); a mesh
    the capsule spanned by its two farthest hull support points with the median distance of the others to that axis."""
    gtype = g.attrib.get("type", defaults.get("type", "sphere"))
    size = _floats(g.attrib.get("size"), [0.0])
    if gtype == "sphere":
        c = _floats(g.attrib.get("pos"), [0, 0, 0])
        return [np.concatenate([c, c, [size[0]]])]
    if gtype in ("capsule", "cylinder"):
        (p0, r), (p1, _) = pts[0], pts[1]
        return [np.concatenate([p0, p1, [r]])]
    if gtype == "box":
        pos = _floats(g.attrib.get("pos"), [0, 0, 0])
        R = _quat_wxyz_to_mat(_floats(g.attrib.get("quat"), [1, 0, 0, 0]))
        h = size[:3]
        k = int(np.argmax(h))
        others = [a for a in range(3) if a != k]
        u, v = (others[0], others[1]) if h[others[0]] >= h[others[1]] else (others[1], others[0])
        if h[u] > 1.5 * h[v]:
            r = float(h[v])
            half = max(h[k] - r, 0.0)
            off = R[:, u] * (h[u] - r)
            return [np.concatenate([pos + sgn * off - R[:, k] * half, pos + sgn * off + R[:, k] * half, [r]]) for sgn in (-1.0, 1.0)]
        r = float(np.mean(np.delete(h, k)))
        half = max(h[k] - r, 0.0)
        return [np.concatenate([pos - R[:, k] * half, pos + R[:, k] * half, [r]])]
    P = np.array([p for p, _ in pts])
    d = np.linalg.norm(P[:, None] - P[None], axis=-1)
    i, j = np.unravel_index(np.argmax(d), d.shape)
    ax = (P[j] - P[i]) / max(d[i, j], 1e-9)
    off = P - P[i]
    rad = float(np.median(np.linalg.norm(off - np.outer(off @ ax, ax), axis=-1)))
    return [np.concatenate([P[i] + ax * rad, P[j] - ax * rad, [rad]]) if d[i, j] > 2 * rad else np.concatenate([0.5 * (P[i] + P[j])] * 2 + [[0.5 * d[i, j]]])]


def compile_mjcf(path):
    """Parse an MJCF humanoid (free root + hinge joints) into the flat model dict."""
    root = ET.parse(path).getroot()
    wb = root.find("worldbody")
    comp = root.find("compiler")
    radian = comp is not None and comp.attrib.get("angle", "degree") == "radian"
    meshdir = os.path.join(os.path.dirname(path), comp.attrib.get("meshdir", "")) if comp is not None else os.path.dirname(path)
    meshes = {}
    if root.find("asset") is not None:
        for me in root.find("asset").findall("mesh"):
            f = os.path.join(meshdir, me.attrib["file"])
            if os.path.exists(f):
                meshes[me.attrib.get("name", os.path.splitext(me.attrib["file"])[0])] = _stl_vertices(f)
    body0 = wb.find("body")
    dflt = root.find("default")
    geom_defaults, joint_defaults = {}, {}
    if dflt is not None:
        if dflt.find("geom") is not None:
            geom_defaults = dict(dflt.find("geom").attrib)
        if dflt.find("joint") is not None:
            joint_defaults = dict(dflt.find("joint").attrib)
    # named default classes (<default class="visual"><geom contype="0" .../></default>, nested classes inherit their parent's)
    geom_classes = {}

    def walk_defaults(node, inherited):
        for d in node.findall("default"):
            attrs = dict(inherited)
            if d.find("geom") is not None:
                attrs.update(d.find("geom").attrib)
            if "class" in d.attrib:
                geom_classes[d.attrib["class"]] = attrs
            walk_defaults(d, attrs)

    if dflt is not None:
        walk_defaults(dflt, geom_defaults)
    gears = {}
    act = root.find("actuator")
    if act is not None:
        for mtr in act.findall("motor"):
            gears[mtr.attrib["joint"]] = float(mtr.attrib.get("gear", "1").split()[0])

    names, parents, local_t, local_q = [], [], [], []
    mass, com, inertia_o = [], [], []
    jtype, dof_start, dof_count = [], [], []
    dof_axis, dof_kp, dof_kd, dof_arm, dof_lo, dof_hi, dof_effort, dof_names = [], [], [], [], [], [], [], []
    cpts = []
    capsules = []
    extra_caps = []            # (body, shape ordinal, capsule[7]) of every collision capsule beyond a body's first
    shape_ordinal = {}         # body -> collision geoms seen so far

    def add(node, parent):
        idx = len(names)
        names.append(node.attrib.get("name"))
        parents.append(parent)
        # np.fromstring(..., dtype=float) then float32 cast, as SkeletonTree.from_mjcf does
        local_t.append(_floats(node.attrib.get("pos"), [0, 0, 0]))
        local_q.append(_floats(node.attrib.get("quat"), [1, 0, 0, 0]))  # rest rotation child-in-parent (wxyz, MJCF order)
        m_tot, mc, parts = 0.0, np.zeros(3), []
        inert = node.find("inertial")
        if inert is not None:
            m_tot = float(inert.attrib["mass"])
            c = _floats(inert.attrib.get("pos"), [0, 0, 0])
            if "fullinertia" in inert.attrib:
                xx, yy, zz, xy, xz, yz = _floats(inert.attrib["fullinertia"])
                Ic = np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])
            else:
                Ic = np.diag(_floats(inert.attrib.get("diaginertia"), [0, 0, 0]))
            if "quat" in inert.attrib:
                R = _quat_wxyz_to_mat(_floats(inert.attrib["quat"]))
                Ic = R @ Ic @ R.T
            mc = m_tot * c
            parts = [(m_tot, c, Ic)]
        capsules.append(np.zeros(7))
        shape_ordinal[idx] = 0
        first_geom = True
        for g in node.findall("geom"):
            if "class" in g.attrib:   # effective attributes: class defaults, then the geom's own
                g = _Attr({**geom_classes.get(g.attrib["class"], {}), **g.attrib})
            if g.attrib.get("contype") == "0" and g.attrib.get("conaffinity") == "0":
                continue  # visual-only geom (robots)
            if g.attrib.get("type") == "mesh" and g.attrib.get("mesh") not in meshes:
                continue
            gm, gc, gI, pts = _geom_props(g, geom_defaults, meshes)
            # collision shapes of the body, in geom order (Isaac Gym enumerates an actor's rigid shapes in this order: the per-shape filter
            # table of humanoid.py:1205-1226 is indexed by it): the first capsule is the body's primary one, the others are EXTRA capsules
            for ci, cap7 in enumerate(_geom_capsules(g, geom_defaults, pts)):
                if first_geom and ci == 0:
                    capsules[idx] = cap7
                else:
                    extra_caps.append((idx, shape_ordinal[idx], cap7))
            first_geom = False
            shape_ordinal[idx] += 1
            for p, r in pts:
                cpts.append((idx, np.asarray(p, dtype=np.float64), float(r)))
            if inert is None:
                m_tot += gm
                mc += gm * gc
                parts.append((gm, gc, gI))
        c = mc / m_tot if m_tot > 0 else np.zeros(3)
        Io = np.zeros((3, 3))
        for gm, gc, gI in parts:
            Io += _shift_inertia(gI, gm, gc)  # about the body origin (joint anchor)
        mass.append(m_tot)
        com.append(c)
        inertia_o.append(Io)
        joints = node.findall("joint")
        if node.find("freejoint") is not None or any(j.attrib.get("type") == "free" for j in joints):
            jtype.append(JOINT_FREE); dof_start.append(len(dof_axis)); dof_count.append(0)
        elif len(joints) == 0:
            jtype.append(JOINT_FIXED); dof_start.append(len(dof_axis)); dof_count.append(0)
        else:
            axes = [_floats(j.attrib.get("axis", joint_defaults.get("axis", "0 0 1"))) for j in joints]
            is_sph = len(joints) == 3 and np.allclose(np.stack(axes), np.eye(3))
            if not is_sph and len(joints) != 1:
                raise NotImplementedError(f"body {names[-1]}: {len(joints)} hinge joints that are not x,y,z")
            jtype.append(JOINT_SPHERICAL if is_sph else JOINT_REVOLUTE)
            dof_start.append(len(dof_axis)); dof_count.append(len(joints))
            for j, ax in zip(joints, axes):
                get = lambda k, d: float(j.attrib.get(k, joint_defaults.get(k, d)))
                rng = _floats(j.attrib.get("range", joint_defaults.get("range", "0 0")))
                dof_axis.append(ax / np.linalg.norm(ax))
                dof_kp.append(get("stiffness", 0.0)); dof_kd.append(get("damping", 0.0)); dof_arm.append(get("armature", 0.0))
                dof_lo.append(rng[0] if radian else np.deg2rad(rng[0]))  # MJCF default angle unit: degree
                dof_hi.append(rng[1] if radian else np.deg2rad(rng[1]))
                dof_effort.append(gears.get(j.attrib.get("name"), 0.0))
                dof_names.append(j.attrib.get("name"))
        for child in node.findall("body"):
            add(child, idx)

    add(body0, -1)
    nb = len(names)
    level = [0] * nb
    for i in range(1, nb):
        level[i] = level[parents[i]] + 1
    cpts.sort(key=lambda t: t[0])
    extra_caps.sort(key=lambda t: t[0])
    model = {
        "source": os.path.basename(path),
        "body_names": names,
        "parent": parents,
        "level": level,
        "local_translation": np.array(local_t, dtype=np.float32).tolist(),
        "local_rotation": np.array(local_q, dtype=np.float64).tolist(),
        "mass": mass,
        "com": np.array(com).tolist(),
        "inertia_origin": np.array(inertia_o).tolist(),
        "joint_type": jtype,
        "dof_start": dof_start,
        "dof_count": dof_count,
        "dof_names": dof_names,
        "dof_axis": np.array(dof_axis).tolist(),
        "dof_kp": dof_kp, "dof_kd": dof_kd, "dof_armature": dof_arm,
        "dof_lower": dof_lo, "dof_upper": dof_hi, "dof_effort": dof_effort,
        "contact_body": [c[0] for c in cpts],
        "contact_pos": [c[1].tolist() for c in cpts],
        "contact_radius": [c[2] for c in cpts],
        "collision_capsule": np.array(capsules).tolist(),
        # [body, shape ordinal within the body (geom order), a[3], b[3], radius] of every collision capsule beyond a body's first
        "extra_capsules": [[int(b), int(o)] + np.asarray(c7, dtype=np.float64).tolist() for b, o, c7 in extra_caps],
        "shapes_per_body": [int(shape_ordinal[i]) for i in range(nb)],
    }
    return model


class ArticulationModel:
    """Numpy view of a compiled model + the packed fp32 / int32 buffers the kernels read."""

    MAX_BODIES = 64   # PHC_MAX_BODIES: slots per model table

    def __init__(self, d):
        self.d = d
        self.body_names = list(d["body_names"])
        self.num_bodies = len(self.body_names)
        assert self.num_bodies <= self.MAX_BODIES, "the stepper maps one body per lane of a 32- or 64-lane group"
        self.parent = np.array(d["parent"], dtype=np.int32)
        self.level = np.array(d["level"], dtype=np.int32)
        self.local_translation = np.array(d["local_translation"], dtype=np.float32)
        lq = np.array(d.get("local_rotation", [[1.0, 0, 0, 0]] * self.num_bodies), dtype=np.float64)
        self.local_rotation = lq / np.linalg.norm(lq, axis=-1, keepdims=True)   # wxyz
        self.mass = np.array(d["mass"], dtype=np.float64)
        self.com = np.array(d["com"], dtype=np.float64)
        self.inertia_origin = np.array(d["inertia_origin"], dtype=np.float64)
        self.joint_type = np.array(d["joint_type"], dtype=np.int32)
        self.dof_start = np.array(d["dof_start"], dtype=np.int32)
        self.dof_count = np.array(d["dof_count"], dtype=np.int32)
        self.dof_names = list(d["dof_names"])
        self.num_dof = len(self.dof_names)
        self.dof_axis = np.array(d["dof_axis"], dtype=np.float64).reshape(-1, 3)
        self.dof_kp = np.array(d["dof_kp"], dtype=np.float64)
        self.dof_kd = np.array(d["dof_kd"], dtype=np.float64)
        self.dof_armature = np.array(d["dof_armature"], dtype=np.float64)
        self.dof_lower = np.array(d["dof_lower"], dtype=np.float64)
        self.dof_upper = np.array(d["dof_upper"], dtype=np.float64)
        self.dof_effort = np.array(d["dof_effort"], dtype=np.float64)
        self.contact_body = np.array(d["contact_body"], dtype=np.int32)
        self.contact_pos = np.array(d["contact_pos"], dtype=np.float64).reshape(-1, 3)
        self.contact_radius = np.array(d["contact_radius"], dtype=np.float64)
        self.collision_capsule = np.array(d.get("collision_capsule", np.zeros((self.num_bodies, 7))), dtype=np.float64).reshape(-1, 7)
        # collision SHAPES (round 4): shape s < NB is body s's primary capsule; shapes NB.. are the extra capsules (second half of a flat box,
        # further geoms of a multi-geom link), each owned by a body.  `shape_filter` = Isaac Gym's per-shape filter word (humanoid.py:1205-1226).
        ex = np.array(d.get("extra_capsules", []), dtype=np.float64).reshape(-1, 9)
        self.extra_owner = ex[:, 0].astype(np.int32)
        self.extra_ordinal = ex[:, 1].astype(np.int32)
        self.extra_capsule = ex[:, 2:9].copy()
        self.shapes_per_body = np.array(d.get("shapes_per_body", [1 if c[6] > 0 else 0 for c in self.collision_capsule]), dtype=np.int32)
        self.shape_filter = np.zeros(self.num_bodies + len(self.extra_owner), dtype=np.int64)
        self.collision_filter = np.zeros(self.num_bodies, dtype=np.int64)   # Isaac Gym shape filter bits (robots.py installs them)
        self.max_level = int(self.level.max())
        self.all_spherical = bool(np.all(self.joint_type[1:] == JOINT_SPHERICAL))
        self.all_revolute = bool(np.all(self.joint_type[1:] == JOINT_REVOLUTE))

    # ---- lookups mirroring the reference task helpers -------------------------------------
    def body_ids(self, names):
        """humanoid.py:1674-1691 `_build_key_body_ids_tensor`."""
        return np.array([self.body_names.index(n) for n in names], dtype=np.int64)

    @property
    def total_mass(self):
        return float(self.mass.sum())

    def children(self):
        ch = [[] for _ in range(self.num_bodies)]
        for i in range(1, self.num_bodies):
            ch[self.parent[i]].append(i)
        return ch

    def solver_tree(self):
        """The tree the stepper's backward / acceleration sweeps walk.  A free-floating articulation can be solved from ANY body as the
        floating base; a level-step of the sweeps is issued once per tree level whatever the number of bodies at that level, so the base
        that minimises the depth is the cheapest: the SMPL humanoid is 8 deep from the pelvis (pelvis ... hand) and 6 from `Spine`
        (arms and legs balance).  State, kinematics, integration and every published tensor stay pelvis-rooted; only the solve re-roots.

        Bodies on the path old root -> base are REVERSED: their solver parent is their child on the path, the joint that connects them
        to it is that child's joint, and their spatial quantities are taken about that joint's anchor (`s_off`, body frame) instead of
        their own origin, so that the joint keeps the motion subspace [1; 0].  All-spherical articulations only (robots keep the root).

        -> dict(base, sparent, slevel, schildren, s_off [NB,3], jsrc (body whose joint links b to its solver parent; -1 base),
                bsrc (path bodies: the reversed body that solves their joint; else -1))"""
        nb = self.num_bodies
        ident = dict(base=0, sparent=self.parent.copy(), slevel=self.level.copy(), schildren=self.children(), s_off=np.zeros((nb, 3)),
                     jsrc=np.array([-1] + list(range(1, nb))), bsrc=np.full(nb, -1))
        if not self.all_spherical or nb < 3 or not getattr(self, "reroot", True):
            return ident
        adj = self.children()
        for i in range(1, nb):
            adj[i] = adj[i] + [int(self.parent[i])]

        def levels_from(base):
            lv, par, order = np.full(nb, -1), np.full(nb, -1), [base]
            lv[base] = 0
            for b in order:
                for c in adj[b]:
                    if lv[c] < 0:
                        lv[c], par[c] = lv[b] + 1, b
                        order.append(c)
            return lv, par
        best, best_depth = 0, self.max_level
        for b in range(nb):
            lv, par = levels_from(b)
            kids = np.bincount(par[par >= 0], minlength=nb)
            if lv.max() < best_depth and kids.max() <= 3:
                best, best_depth = b, int(lv.max())
        if best == 0:
            return ident
        lv, par = levels_from(best)
        sch = [[] for _ in range(nb)]
        for i in range(nb):
            if par[i] >= 0:
                sch[par[i]].append(i)
        s_off, jsrc, bsrc = np.zeros((nb, 3)), np.arange(nb), np.full(nb, -1)
        jsrc[best] = -1
        c = best
        while c != 0:                       # walk the path base -> old root: p is reversed, its solver joint is c's
            p = int(self.parent[c])
            assert par[p] == c
            s_off[p] = self.local_translation[c]
            jsrc[p] = c
            bsrc[c] = p
            c = p
        return dict(base=best, sparent=par, slevel=lv, schildren=sch, s_off=s_off, jsrc=jsrc, bsrc=bsrc)

    @property
    def num_shapes_collision(self):
        return self.num_bodies + len(self.extra_owner)

    def shape_owner(self):
        return np.concatenate([np.arange(self.num_bodies, dtype=np.int32), self.extra_owner])

    def shape_capsules(self):
        return np.concatenate([self.collision_capsule, self.extra_capsule.reshape(-1, 7)], axis=0)

    def set_body_filters(self, body_filters):
        """One Isaac Gym filter word per BODY (every shape of the body takes it)."""
        f = np.asarray(body_filters, dtype=np.int64)
        self.collision_filter[:] = f
        self.shape_filter = f[self.shape_owner()].copy()

    def set_shape_filters(self, shape_filters):
        """One filter word per SHAPE in the reference's order (bodies in order, a body's shapes in geom order; humanoid.py:1205-1226).  A body's
        capsules beyond its geoms (the second half of a split box) share their geom's word."""
        f = np.asarray(shape_filters, dtype=np.int64)
        assert len(f) == int(self.shapes_per_body.sum()), (len(f), int(self.shapes_per_body.sum()))
        start = np.concatenate([[0], np.cumsum(self.shapes_per_body)[:-1]])
        self.shape_filter = np.zeros(self.num_shapes_collision, dtype=np.int64)
        for b in range(self.num_bodies):
            n = int(self.shapes_per_body[b])
            if n:
                self.shape_filter[b] = f[start[b]]
                self.collision_filter[b] = int(np.bitwise_or.reduce(f[start[b]:start[b] + n]))
        for e, (b, o) in enumerate(zip(self.extra_owner, self.extra_ordinal)):
            self.shape_filter[self.num_bodies + e] = f[start[b] + min(int(o), int(self.shapes_per_body[b]) - 1)]

    def collision_allow_masks(self):
        """Bit j of entry i: bodies i and j may collide -- not the same body, not joined by a joint (PhysX articulations never
        collide parent and child), no common Isaac Gym filter bit on their PRIMARY shapes (humanoid.py:1205-1226: shapes collide iff
        (fa & fb) == 0), and both carry a collision capsule.  (Informative: the kernels walk `collision_pairs()`.)"""
        nb = self.num_bodies
        has = self.collision_capsule[:, 6] > 0
        sf = self.shape_filter if len(self.shape_filter) >= nb and self.shape_filter.any() else self.collision_filter
        m = [0] * nb   # Python ints: bit 63 does not fit a signed 64-bit numpy scalar
        for i in range(nb):
            for j in range(nb):
                if i != j and has[i] and has[j] and self.parent[i] != j and self.parent[j] != i and (int(sf[i]) & int(sf[j])) == 0:
                    m[i] |= 1 << j
        return m

    def collision_pairs(self):
        """Candidate pairs of collision SHAPES (s1 < s2): different owner bodies that are not joined by a joint, no common filter bit, both with
        a radius.  With one capsule per body this is the body-pair list of rounds 1-3."""
        own, cap = self.shape_owner(), self.shape_capsules()
        sf = self.shape_filter if self.shape_filter.any() else self.collision_filter[own]
        ns = len(own)
        out = []
        for a in range(ns):
            for b in range(a + 1, ns):
                i, j = int(own[a]), int(own[b])
                if i != j and cap[a, 6] > 0 and cap[b, 6] > 0 and self.parent[i] != j and self.parent[j] != i and (int(sf[a]) & int(sf[b])) == 0:
                    out.append((a, b))
        return out

    # ---- packed buffers --------------------------------------------------------------------
    def pack(self, kp_scale=1.0, kd_scale=1.0, float_dtype=np.float32):
        """-> (ints int32[...], floats float32[...]) laid out as csrc/phc_model.h expects.

        ints  : [0]=NB [1]=ND [2]=max_level [3]=NCP then per body (MAX_BODIES slots each):
                parent, level, joint_type, dof_start, child0, child1, child2, nchild,
                cp_start, cp_count, order (bodies sorted by level), misc [-, -, solver depth, solver base, jump steps],
                self-collision partner mask
        floats: per body (MAX_BODIES slots x BODY_FLOATS): r_local[3], mass, m*com[3], Io(xx,xy,xz,yy,yz,zz)[6],
                kp[3], kd[3], armature[3], effort[3], axis[3] (revolute), rest rotation xyzw[4], limit lo/hi [2], contact bound radius, pad, collision capsule a[3] b[3] radius, pad,
                solver reference point s_off[3], m*(com - s_off)[3], inertia about it [6] ;
                then contact points NCP x 4 (pos[3], radius)
        """
        NB, MB = self.num_bodies, self.MAX_BODIES
        NT = 21
        ints = np.zeros(4 + NT * MB, dtype=np.int32)
        ints[0:4] = [NB, self.num_dof, self.max_level, len(self.contact_body)]
        tab = ints[4:].reshape(NT, MB)
        tab[0, :] = -1
        tab[1, :] = -1
        tab[4:7, :] = -1
        ch = self.children()
        for i in range(NB):
            assert len(ch[i]) <= 3, "at most 3 children per body supported by the sweep"
            tab[0, i] = self.parent[i]
            tab[1, i] = self.level[i]
            tab[2, i] = self.joint_type[i]
            tab[3, i] = self.dof_start[i]
            for k, c in enumerate(ch[i]):
                tab[4 + k, i] = c
            tab[7, i] = len(ch[i])
            idx = np.nonzero(self.contact_body == i)[0]
            tab[8, i] = idx[0] if len(idx) else 0
            tab[9, i] = len(idx)
        tab[10, :NB] = np.argsort(self.level, kind="stable")  # bodies sorted by tree level
        # misc table 11: [0, 0 (reserved), solver depth, solver base, pointer-jumping steps]
        tab[12, :NB] = np.array([v & 0xffffffff for v in self.collision_allow_masks()], dtype=np.uint32).view(np.int32)   # partner bit masks (bodies 0-31; informative)
        # solver tree (solver_tree()): tables 13..19 = parent, level, child0-2, nchild, jsrc | bsrc << 8 (+1 each, 0 = none); misc[2:4] = its depth, its base
        st = self.solver_tree()
        tab[13, :] = -1
        tab[14, :] = -1
        tab[15:18, :] = -1
        for i in range(NB):
            assert len(st["schildren"][i]) <= 3
            tab[13, i] = st["sparent"][i]
            tab[14, i] = st["slevel"][i]
            for k, c in enumerate(st["schildren"][i]):
                tab[15 + k, i] = c
            tab[18, i] = len(st["schildren"][i])
            tab[19, i] = (int(st["jsrc"][i]) + 1) | ((int(st["bsrc"][i]) + 1) << 8)
        tab[11, 2:4] = [int(st["slevel"].max()), st["base"]]
        # kinematics by pointer jumping (phc_aba.h aba_fk_jump_*): table 20 = the body's anchor in steps 0..3 (8 bits each, 0xff = already in the
        # world frame): step k composes the body's transform with its anchor's, the anchor's anchor becomes the new anchor; misc[4] = steps
        anc = [int(p) for p in self.parent[:NB]]
        steps, words = 0, [0xffffffff] * NB
        while any(a >= 0 for a in anc):
            assert steps < 4, "kinematic tree deeper than 15"
            for i in range(NB):
                words[i] = (words[i] & ~(0xff << (8 * steps))) | ((anc[i] & 0xff) << (8 * steps))
            anc = [anc[a] if a >= 0 else -1 for a in anc]
            steps += 1
        tab[20, :] = -1
        tab[20, :NB] = np.array(words, dtype=np.uint32).view(np.int32)
        tab[11, 4] = steps
        # self-collision candidate pairs (i < k, may collide), appended after the tables: [count, i | k << 8, ...]; lane l of a
        # group evaluates pairs l, l + L, l + 2L, ... so the list is ordered to spread each body's pairs over many lanes
        pairs = self.collision_pairs()
        # lanes work through the list in lock step (pair t * L + l in round t): sorted by the gap between the bounding spheres in
        # the rest pose, a round holds pairs that are either all near (everyone runs the capsule test) or all far (nobody does)
        org = np.zeros((NB, 3))
        for j in range(1, NB):
            org[j] = org[self.parent[j]] + self.local_translation[j]
        own, cap = self.shape_owner(), self.shape_capsules()
        NX = len(self.extra_owner)
        assert NB + NX <= (32 if NB <= 32 else MB), "the collision shapes of an env must fit the lanes of its group"
        tab[11, 5] = NX
        mid = org[own] + 0.5 * (cap[:, 0:3] + cap[:, 3:6])
        rad = 0.5 * np.linalg.norm(cap[:, 3:6] - cap[:, 0:3], axis=-1) + cap[:, 6]
        pairs.sort(key=lambda ik: np.linalg.norm(mid[ik[0]] - mid[ik[1]]) - rad[ik[0]] - rad[ik[1]])
        ints = np.concatenate([ints, np.array([len(pairs)] + [i | (k << 8) for i, k in pairs], dtype=np.int32)])
        BF = self.BODY_FLOATS
        fl = np.zeros((MB, BF), dtype=np.float64)
        fl[:, 31] = 1.0  # identity rest rotation (xyzw)
        for i in range(NB):
            fl[i, 0:3] = self.local_translation[i]
            fl[i, 3] = self.mass[i]
            fl[i, 4:7] = self.mass[i] * self.com[i]
            I = self.inertia_origin[i]
            fl[i, 7:13] = [I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]]
            s, c = self.dof_start[i], self.dof_count[i]
            if c > 0:
                fl[i, 13:13 + c] = self.dof_kp[s:s + c] * kp_scale
                fl[i, 16:16 + c] = self.dof_kd[s:s + c] * kd_scale
                fl[i, 19:19 + c] = self.dof_armature[s:s + c]
                fl[i, 22:22 + c] = self.dof_effort[s:s + c]
                if self.joint_type[i] == JOINT_REVOLUTE:
                    fl[i, 25:28] = self.dof_axis[s]
                    lo, hi = self.dof_limits()
                    fl[i, 32], fl[i, 33] = lo[s], hi[s]
            w, x, y, z = self.local_rotation[i]
            fl[i, 28:32] = [x, y, z, w]
            idx = np.nonzero(self.contact_body == i)[0]
            # contact broad phase: no point of the body can touch z = 0 while the body origin is higher than this
            fl[i, 34] = (np.linalg.norm(self.contact_pos[idx], axis=-1) + self.contact_radius[idx]).max() * 1.0001 if len(idx) else 0.0
            fl[i, 36:43] = self.collision_capsule[i]
            # solver reference point (solver_tree()): offset, m * com and inertia about it (= the origin's for every body that is not reversed)
            so = st["s_off"][i]
            cr = self.com[i] - so
            Icom = I - self.mass[i] * (self.com[i] @ self.com[i] * np.eye(3) - np.outer(self.com[i], self.com[i]))
            Ir = Icom + self.mass[i] * (cr @ cr * np.eye(3) - np.outer(cr, cr))
            fl[i, 44:47] = so
            fl[i, 47:50] = self.mass[i] * cr
            fl[i, 50:56] = [Ir[0, 0], Ir[0, 1], Ir[0, 2], Ir[1, 1], Ir[1, 2], Ir[2, 2]]
        cp = np.concatenate([self.contact_pos, self.contact_radius[:, None]], axis=1) if len(self.contact_body) else np.zeros((0, 4))
        # within a body: lowest points (local z - radius) first.  The stepper walks a body's points in lockstep over the lanes of a
        # wavefront and skips an iteration's force / inertia part when NO lane's point touches the ground: with the soles' points at the
        # front of every foot's list, the iterations of the upper box corners are skipped together (a box came as z-alternating corners).
        for i in range(NB):
            idx = np.nonzero(self.contact_body == i)[0]
            if len(idx) > 1:
                assert (np.diff(idx) == 1).all()
                order = np.argsort(cp[idx, 2] - cp[idx, 3], kind="stable")
                cp[idx] = cp[idx][order]
        # extra collision capsules after the contact points: a[3], b[3], radius, owner body (as a float) each
        xc = np.concatenate([self.extra_capsule.reshape(-1, 7), self.extra_owner.reshape(-1, 1).astype(np.float64)], axis=1) if NX else np.zeros((0, 8))
        floats = np.concatenate([fl.reshape(-1), cp.reshape(-1), xc.reshape(-1)]).astype(float_dtype)
        return ints, floats

    BODY_FLOATS = 56

    # ---- action scaling (A1) ---------------------------------------------------------------
    def dof_limits(self):
        """humanoid.py:966-983: swap if lower>upper; equal & zero -> +-pi."""
        lo, hi = [], []
        for l, u in zip(self.dof_lower, self.dof_upper):
            l, u = np.float32(l), np.float32(u)
            if l > u:
                lo.append(u); hi.append(l)
            elif l == u:
                lo.append(-np.pi); hi.append(np.pi)
            else:
                lo.append(l); hi.append(u)
        return np.array(lo, dtype=np.float32), np.array(hi, dtype=np.float32)

    def pd_action_offset_scale(self, bias_offset=False, has_smpl_pd_offset=False, has_upright_start=True):
        """Reference `_build_pd_action_offset_scale` (humanoid.py:1331-1409); pinned to the reference's own method by
        tests/golden/pd_offset_scale.npz (oracle/gen_golden_learner.py)."""
        lim_low, lim_high = self.dof_limits()
        lim_low, lim_high = lim_low.copy(), lim_high.copy()
        for i in range(1, self.num_bodies):
            s, c = self.dof_start[i], self.dof_count[i]
            if c == 0:
                continue
            if not bias_offset:
                if c == 3:
                    cl = np.max(np.abs(lim_low[s:s + 3]))
                    chh = np.max(np.abs(lim_high[s:s + 3]))
                    sc = min([1.2 * max([cl, chh]), np.pi])
                    lim_low[s:s + 3] = -sc
                    lim_high[s:s + 3] = sc
                elif c == 1:
                    mid = 0.5 * (lim_high[s] + lim_low[s])
                    sc = 0.7 * (lim_high[s] - lim_low[s])
                    lim_low[s] = mid - sc
                    lim_high[s] = mid + sc
            else:
                mid = 0.5 * (lim_high[s:s + c] + lim_low[s:s + c])
                sc = 0.7 * (lim_high[s:s + c] - lim_low[s:s + c])
                lim_low[s:s + c] = mid - sc
                lim_high[s:s + c] = mid + sc
        offset = (0.5 * (lim_high + lim_low)).astype(np.float32)
        scale = (0.5 * (lim_high - lim_low)).astype(np.float32)
        if self.all_spherical and "L_Knee" in self.body_names:
            jn = self.body_names[1:]
            scale[jn.index("L_Knee") * 3 + 1] = 5  # "Bumping Kneel" humanoid.py:1387-1394
            scale[jn.index("R_Knee") * 3 + 1] = 5
            if has_smpl_pd_offset:   # humanoid.py:1396-1404
                ls, rs = jn.index("L_Shoulder") * 3, jn.index("R_Shoulder") * 3
                if has_upright_start:
                    offset[ls], offset[rs] = -np.pi / 2, np.pi / 2
                else:
                    offset[ls], offset[ls + 2] = -np.pi / 6, -np.pi / 2
                    offset[rs], offset[rs + 2] = -np.pi / 3, np.pi / 2
        elif not self.all_spherical:
            offset[:] = 0            # humanoid.py:1406-1407 (h1 / g1)
        return offset, scale

    def limb_lengths_and_weights(self, groups):
        """humanoid.py:1097-1106: per-group summed |local_translation| and masses."""
        ll = np.linalg.norm(self.local_translation, axis=-1)
        out = [ll[g].sum() for g in groups] + [self.mass[g].sum() for g in groups]
        return np.array(out, dtype=np.float32)


def pack_shapes(models, kp_scale=1.0, kd_scale=1.0):
    """K articulations of ONE topology (per-env body shapes, humanoid.py:726-766,824-866) -> (ints [K, ni], floats [K, nf]): their
    packed tables side by side, zero-padded to a common length (contact-point and collision-pair counts may differ)."""
    m0 = models[0]
    for m in models[1:]:
        if (m.body_names != m0.body_names or not np.array_equal(m.parent, m0.parent) or not np.array_equal(m.joint_type, m0.joint_type)
                or not np.array_equal(m.dof_start, m0.dof_start) or m.num_dof != m0.num_dof):
            raise ValueError("shape variants must share body names, parents and joint layout")
    packed = [m.pack(kp_scale, kd_scale) for m in models]
    ni, nf = max(p[0].size for p in packed), max(p[1].size for p in packed)
    ints = np.zeros((len(models), ni), dtype=np.int32)
    floats = np.zeros((len(models), nf), dtype=np.float32)
    for k, (i, f) in enumerate(packed):
        ints[k, :i.size], floats[k, :f.size] = i, f
    return ints, floats


def load_model(name_or_path):
    """Load a compiled JSON asset by name ("smpl_humanoid") or path, or compile an MJCF."""
    p = name_or_path
    if not os.path.exists(p):
        p = os.path.join(ASSET_DIR, name_or_path if name_or_path.endswith(".json") else name_or_path + ".json")
    if p.endswith(".xml"):
        return ArticulationModel(compile_mjcf(p))
    with open(p) as f:
        return ArticulationModel(json.load(f))


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    m = compile_mjcf(src)
    with open(dst, "w") as f:
        json.dump(m, f, indent=0)
    am = ArticulationModel(m)
    print(f"{src}: {am.num_bodies} bodies, {am.num_dof} dof, mass {am.total_mass:.2f} kg, "
          f"{len(am.contact_body)} contact points, max level {am.max_level}")
