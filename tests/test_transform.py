"""Hand-derived known answers for the camera conventions both the product and the oracle consume
(``_transform.py``; in the reference these matrices come from pygfx / pylinalg, absent here).  The parity
tests feed the SAME matrices to both sides, so a wrong convention would be invisible to them: these pins
are what stands in for that."""
import numpy as np

from sub_volume_renderer_amd._transform import AffineTransform, PerspectiveCamera


def test_projection_matrix_fov_applies_to_the_mean_extent_and_depth_maps_to_0_1():
    # pygfx PerspectiveCamera: the field of view spans the MEAN of the view's width and height.
    # fov 90 deg, aspect 2, near 1, far 3:  mean extent at the near plane = 2 * near * tan(45 deg) = 2
    #   height = 2 * 2 / (1 + aspect) = 4/3,  width = aspect * height = 8/3  (mean = 2)
    #   right = 4/3, top = 2/3
    cam = PerspectiveCamera(90.0, 2.0, depth_range=(1.0, 3.0))
    want = np.array([[1.0 / (4.0 / 3.0), 0, 0, 0],
                     [0, 1.0 / (2.0 / 3.0), 0, 0],
                     [0, 0, 3.0 / (1.0 - 3.0), 1.0 * 3.0 / (1.0 - 3.0)],
                     [0, 0, -1.0, 0]])
    np.testing.assert_allclose(cam.projection_matrix, want, rtol=0, atol=1e-12)
    P = cam.projection_matrix
    # the near plane's top-right corner -> NDC (1, 1, 0); a far-plane point on the axis -> z = 1 (wgpu depth [0, 1])
    c = P @ np.array([4.0 / 3.0, 2.0 / 3.0, -1.0, 1.0])
    np.testing.assert_allclose(c[:3] / c[3], [1.0, 1.0, 0.0], atol=1e-12)
    c = P @ np.array([0.0, 0.0, -3.0, 1.0])
    np.testing.assert_allclose(c[2] / c[3], 1.0, atol=1e-12)
    np.testing.assert_allclose(cam.projection_matrix_inverse @ P, np.eye(4), atol=1e-12)


def test_projection_matrix_square_view_is_the_textbook_frustum():
    cam = PerspectiveCamera(60.0, 1.0, depth_range=(0.5, 100.0))
    f = 1.0 / np.tan(np.radians(30.0))
    P = cam.projection_matrix
    np.testing.assert_allclose([P[0, 0], P[1, 1]], [f, f], rtol=1e-12)
    np.testing.assert_allclose(P[2, 2], 100.0 / (0.5 - 100.0))
    np.testing.assert_allclose(P[2, 3], 0.5 * 100.0 / (0.5 - 100.0))
    cam.zoom = 2.0                                        # zoom narrows the extent
    np.testing.assert_allclose(cam.projection_matrix[0, 0], 2 * f, rtol=1e-12)


def test_look_at_points_local_minus_z_at_the_target_with_y_up():
    cam = PerspectiveCamera(45.0, 1.0)
    cam.world.position = (1.0, 2.0, 3.0)
    cam.look_at((1.0, 2.0, -5.0))                         # straight down -z: no rotation at all
    np.testing.assert_allclose(cam.world.rotation_matrix, np.eye(3), atol=1e-12)
    cam.look_at((11.0, 2.0, 3.0))                         # down +x: local -z = +x, local y stays +y, local x = +z... right-handed
    R = cam.world.rotation_matrix
    np.testing.assert_allclose(R @ np.array([0, 0, -1.0]), [1, 0, 0], atol=1e-12)
    np.testing.assert_allclose(R @ np.array([0, 1.0, 0]), [0, 1, 0], atol=1e-12)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-12)
    # view matrix = inverse of the camera's world matrix: the eye maps to the origin, the target onto -z
    V = cam.view_matrix
    np.testing.assert_allclose(V @ np.array([1.0, 2.0, 3.0, 1.0]), [0, 0, 0, 1], atol=1e-12)
    np.testing.assert_allclose(V @ np.array([11.0, 2.0, 3.0, 1.0]), [0, 0, -10, 1], atol=1e-12)


def test_affine_transform_composes_translate_rotate_scale():
    t = AffineTransform()
    t.position = (1.0, 2.0, 3.0)
    t.scale = (1.0, 1.0, 6.0)                             # scripts/mouse.py:90-91 (world.scale_z = 6)
    np.testing.assert_allclose(t.matrix @ np.array([1.0, 1.0, 1.0, 1.0]), [2, 3, 9, 1])
    np.testing.assert_allclose(t.inverse_matrix @ t.matrix, np.eye(4), atol=1e-12)
    assert t.scale_z == 6.0
