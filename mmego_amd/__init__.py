"""mmego_amd: the mmEgo hot path (IMU_Net -> Upper_Net -> Lower_Net forward/backward, L1 loss, Adam) as
hand-written HIP kernels for MI355X (gfx950), behind the reference's nn.Module / CLI surface.

Importing the package never touches the GPU; using a net does, and fails loudly without the HIP library.
"""
__all__ = ["hip", "ops", "nets", "params", "skeleton", "build"]
