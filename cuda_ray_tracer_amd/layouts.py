"""numpy dtypes matching the POD structs of include/mirt.h (and so the reference's classes, object.cuh)."""
import numpy as np

MAT = np.dtype([("color", "<f4", 3), ("shininess", "<f4", 3), ("trans", "<f4", 3), ("ior", "<f4"), ("roughness", "<f4")])
SPHERE = np.dtype([("c", "<f4", 3), ("r", "<f4"), ("mat", MAT)])
TRIANGLE = np.dtype([("p0", "<f4", 3), ("p1", "<f4", 3), ("p2", "<f4", 3), ("nor", "<f4", 3), ("e1", "<f4", 3), ("e2", "<f4", 3), ("mat", MAT)])
PLANE = np.dtype([("abcd", "<f4", 4), ("nor", "<f4", 3), ("point", "<f4", 3), ("mat", MAT)])
LIGHT = np.dtype([("v", "<f4", 3), ("color", "<f4", 3)])
PRIMREF = np.dtype([("type", "<u4"), ("id", "<u4")])
TREENODE = np.dtype([("xmin", "<f4"), ("xmax", "<f4"), ("ymin", "<f4"), ("ymax", "<f4"), ("zmin", "<f4"), ("zmax", "<f4"),
                     ("left", "<u4"), ("right", "<u4"), ("prim_offset", "<u4"), ("count", "<u4")])
assert (MAT.itemsize, SPHERE.itemsize, TRIANGLE.itemsize, PLANE.itemsize, LIGHT.itemsize, PRIMREF.itemsize) == (44, 60, 116, 84, 24, 8)
