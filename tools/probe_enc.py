import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import torch
mic = entry.load_package()
img = np.fromfile("tests/golden/MR_256_256_image.bin", dtype="<u2").reshape(256, 256)
d_px = torch.from_numpy(img.view(np.int16)).cuda()
for ns in (2, 4, 8):
    sess = mic.Session(1, 65536); cu = mic.Session.make_units([(0, 256, 256, int(img.max()), ns)])
    sess.encode_enqueue(d_px.data_ptr(), cu); d_blobs, offs, st, used = sess.encode_finish()
    buf = (C.c_uint32 * 16)(); mic.lib().mic_hip_debug_unit(sess._h, 0, buf)
    print(ns, "status", st, "used", used, "len", offs[-1], "probe rc", C.c_int32(buf[8]).value, "total_bytes", buf[9], "lanes", buf[7], "ntok", buf[0], "hdr", buf[5], "gate rc", C.c_int32(buf[12]).value, "sym_bits", buf[13], "N", buf[14])
