"""camera_sync.v4l2 (the stand-in for cv.VideoCapture(deviceId), reference camera.py:98-118): no camera exists where the
tests run, so what can be checked is the contract with the kernel -- every structure size, field offset, ioctl number and
constant the module uses, against <linux/videodev2.h> through a probe compiled here -- and the behaviour without a device."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import pytest

from cameracalibration_amd.camera_sync import v4l2

PROBE = r"""
#include <stdio.h>
#include <stddef.h>
#include <linux/videodev2.h>
#define P(name, v) printf("%s %lu\n", name, (unsigned long)(v))
int main(void) {
    P("sizeof_capability", sizeof(struct v4l2_capability)); P("sizeof_format", sizeof(struct v4l2_format));
    P("sizeof_requestbuffers", sizeof(struct v4l2_requestbuffers)); P("sizeof_buffer", sizeof(struct v4l2_buffer));
    P("sizeof_control", sizeof(struct v4l2_control)); P("sizeof_pix", sizeof(struct v4l2_pix_format));
    P("cap.capabilities", offsetof(struct v4l2_capability, capabilities)); P("cap.device_caps", offsetof(struct v4l2_capability, device_caps));
    P("fmt.pix.width", offsetof(struct v4l2_format, fmt.pix.width)); P("fmt.pix.height", offsetof(struct v4l2_format, fmt.pix.height));
    P("fmt.pix.pixelformat", offsetof(struct v4l2_format, fmt.pix.pixelformat)); P("fmt.pix.field", offsetof(struct v4l2_format, fmt.pix.field));
    P("fmt.pix.sizeimage", offsetof(struct v4l2_format, fmt.pix.sizeimage));
    P("req.memory", offsetof(struct v4l2_requestbuffers, memory));
    P("buf.bytesused", offsetof(struct v4l2_buffer, bytesused)); P("buf.memory", offsetof(struct v4l2_buffer, memory));
    P("buf.m.offset", offsetof(struct v4l2_buffer, m.offset)); P("buf.length", offsetof(struct v4l2_buffer, length));
    P("ctrl.value", offsetof(struct v4l2_control, value));
    P("VIDIOC_QUERYCAP", VIDIOC_QUERYCAP); P("VIDIOC_G_FMT", VIDIOC_G_FMT); P("VIDIOC_S_FMT", VIDIOC_S_FMT);
    P("VIDIOC_REQBUFS", VIDIOC_REQBUFS); P("VIDIOC_QUERYBUF", VIDIOC_QUERYBUF); P("VIDIOC_QBUF", VIDIOC_QBUF);
    P("VIDIOC_DQBUF", VIDIOC_DQBUF); P("VIDIOC_STREAMON", VIDIOC_STREAMON); P("VIDIOC_STREAMOFF", VIDIOC_STREAMOFF);
    P("VIDIOC_S_CTRL", VIDIOC_S_CTRL);
    P("V4L2_PIX_FMT_YUYV", V4L2_PIX_FMT_YUYV); P("V4L2_PIX_FMT_MJPEG", V4L2_PIX_FMT_MJPEG);
    P("V4L2_CID_FOCUS_ABSOLUTE", V4L2_CID_FOCUS_ABSOLUTE); P("V4L2_CID_FOCUS_AUTO", V4L2_CID_FOCUS_AUTO);
    P("V4L2_BUF_TYPE_VIDEO_CAPTURE", V4L2_BUF_TYPE_VIDEO_CAPTURE); P("V4L2_MEMORY_MMAP", V4L2_MEMORY_MMAP);
    P("V4L2_FIELD_ANY", V4L2_FIELD_ANY); P("V4L2_CAP_VIDEO_CAPTURE", V4L2_CAP_VIDEO_CAPTURE);
    P("V4L2_CAP_STREAMING", V4L2_CAP_STREAMING); P("V4L2_CAP_DEVICE_CAPS", V4L2_CAP_DEVICE_CAPS);
    return 0;
}
"""


@pytest.mark.skipif(not sys.platform.startswith("linux") or not os.path.exists("/usr/include/linux/videodev2.h") or not shutil.which("gcc"),
                    reason="needs Linux, <linux/videodev2.h> and gcc")
def test_layouts_and_ioctl_numbers_are_the_kernel_headers(tmp_path):
    src = tmp_path / "probe.c"
    src.write_text(PROBE)
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", str(src), "-o", str(exe)])
    want = dict((k, int(v)) for k, v in (line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines()))
    fmt_pix = v4l2.v4l2_format.fmt.offset
    have = {
        "sizeof_capability": C.sizeof(v4l2.v4l2_capability), "sizeof_format": C.sizeof(v4l2.v4l2_format),
        "sizeof_requestbuffers": C.sizeof(v4l2.v4l2_requestbuffers), "sizeof_buffer": C.sizeof(v4l2.v4l2_buffer),
        "sizeof_control": C.sizeof(v4l2.v4l2_control), "sizeof_pix": C.sizeof(v4l2.v4l2_pix_format),
        "cap.capabilities": v4l2.v4l2_capability.capabilities.offset, "cap.device_caps": v4l2.v4l2_capability.device_caps.offset,
        "fmt.pix.width": fmt_pix + v4l2.v4l2_pix_format.width.offset, "fmt.pix.height": fmt_pix + v4l2.v4l2_pix_format.height.offset,
        "fmt.pix.pixelformat": fmt_pix + v4l2.v4l2_pix_format.pixelformat.offset, "fmt.pix.field": fmt_pix + v4l2.v4l2_pix_format.field.offset,
        "fmt.pix.sizeimage": fmt_pix + v4l2.v4l2_pix_format.sizeimage.offset,
        "req.memory": v4l2.v4l2_requestbuffers.memory.offset,
        "buf.bytesused": v4l2.v4l2_buffer.bytesused.offset, "buf.memory": v4l2.v4l2_buffer.memory.offset,
        "buf.m.offset": v4l2.v4l2_buffer.m.offset, "buf.length": v4l2.v4l2_buffer.length.offset,
        "ctrl.value": v4l2.v4l2_control.value.offset,
    }
    for name in ("VIDIOC_QUERYCAP", "VIDIOC_G_FMT", "VIDIOC_S_FMT", "VIDIOC_REQBUFS", "VIDIOC_QUERYBUF", "VIDIOC_QBUF", "VIDIOC_DQBUF",
                 "VIDIOC_STREAMON", "VIDIOC_STREAMOFF", "VIDIOC_S_CTRL", "V4L2_PIX_FMT_YUYV", "V4L2_PIX_FMT_MJPEG",
                 "V4L2_CID_FOCUS_ABSOLUTE", "V4L2_CID_FOCUS_AUTO", "V4L2_BUF_TYPE_VIDEO_CAPTURE", "V4L2_MEMORY_MMAP", "V4L2_FIELD_ANY",
                 "V4L2_CAP_VIDEO_CAPTURE", "V4L2_CAP_STREAMING", "V4L2_CAP_DEVICE_CAPS"):
        have[name] = getattr(v4l2, name)
    assert have == want


def test_without_a_device_it_behaves_like_videocapture():
    cap = v4l2.V4L2Capture(61)  # /dev/video61: not here
    assert not cap.isOpened()
    assert cap.set(v4l2.CAP_PROP_FOCUS, 50) is False
    assert cap.read() == (False, None)
    cap.release()
    assert cap.read() == (False, None)
    # something that can be opened but is no video device
    cap = v4l2.V4L2Capture("/dev/null")
    assert not cap.isOpened() and cap.read() == (False, None)
