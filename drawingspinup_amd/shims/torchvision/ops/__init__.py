from drawingspinup_amd.style.generators import deform_conv2d  # noqa: F401
