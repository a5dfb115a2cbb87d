"""Minimal rosbag (format 2.0) reader for the two topics a FLVIS run consumes -- stereo `sensor_msgs/Image` pairs and
`sensor_msgs/Imu` -- without ROS (harness code: the EuRoC bags the reference's launch files play, `bag/bag.md`,
`launch/EuRoC_MAV/*.launch`, can be fed to the tracker directly).  Only what is needed: record framing, chunks (`none` and
`bz2` compression; `lz4` needs a library that is not in this image), connection records, message-data records, and the ROS1
serialisation of std_msgs/Header, sensor_msgs/Image and sensor_msgs/Imu.  `BagWriter` writes uncompressed bags of the same
message types (used by the tests to build fixtures)."""
import bz2
import struct

import numpy as np

MAGIC = b"#ROSBAG V2.0\n"
OP_MSG_DATA, OP_BAG_HEADER, OP_INDEX_DATA, OP_CHUNK, OP_CHUNK_INFO, OP_CONNECTION = 0x02, 0x03, 0x04, 0x05, 0x06, 0x07


def _fields(hdr):
    out, p = {}, 0
    while p < len(hdr):
        (n,) = struct.unpack_from("<I", hdr, p)
        p += 4
        k, _, v = hdr[p:p + n].partition(b"=")
        out[k.decode()] = v
        p += n
    return out


def _records(buf, p=0, end=None):
    end = len(buf) if end is None else end
    while p + 8 <= end:
        (hl,) = struct.unpack_from("<I", buf, p)
        hdr = buf[p + 4:p + 4 + hl]
        (dl,) = struct.unpack_from("<I", buf, p + 4 + hl)
        d0 = p + 8 + hl
        yield _fields(hdr), buf[d0:d0 + dl]
        p = d0 + dl


def _string(buf, p):
    (n,) = struct.unpack_from("<I", buf, p)
    return buf[p + 4:p + 4 + n].decode(errors="replace"), p + 4 + n


def parse_header(buf, p=0):
    """std_msgs/Header -> (seq, stamp_seconds, frame_id, next offset)"""
    seq, secs, nsecs = struct.unpack_from("<III", buf, p)
    frame_id, p = _string(buf, p + 12)
    return seq, secs + nsecs * 1e-9, frame_id, p


def parse_image(buf):
    """sensor_msgs/Image -> (stamp, array [h, w] or [h, w, c] (uint8; uint16 for 16UC1 / mono16), encoding)"""
    _, stamp, _, p = parse_header(buf)
    h, w = struct.unpack_from("<II", buf, p)
    enc, p = _string(buf, p + 8)
    _, step = struct.unpack_from("<BI", buf, p)
    (n,) = struct.unpack_from("<I", buf, p + 5)
    data = np.frombuffer(buf, np.uint8, n, p + 9)
    if enc in ("16UC1", "mono16"):
        img = data.reshape(h, step)[:, :2 * w].copy().view(np.uint16)
    else:
        ch = {"mono8": 1, "8UC1": 1, "bgr8": 3, "rgb8": 3, "bgra8": 4, "rgba8": 4}.get(enc)
        if ch is None:
            raise ValueError("unsupported image encoding %r" % enc)
        img = data.reshape(h, step)[:, :w * ch].copy()
        if ch > 1:
            img = img.reshape(h, w, ch)
    return stamp, img, enc


def parse_imu(buf):
    """sensor_msgs/Imu -> (stamp, gyro xyz, acc xyz) in the sensor frame"""
    _, stamp, _, p = parse_header(buf)
    v = struct.unpack_from("<37d", buf, p)  # orientation 4, cov 9, angular_velocity 3, cov 9, linear_acceleration 3, cov 9
    return stamp, np.array(v[13:16]), np.array(v[25:28])


class BagReader:
    """Iterates the (topic, datatype, receive time, raw message bytes) of a bag in file order."""

    def __init__(self, path):
        self.buf = open(path, "rb").read()
        if not self.buf.startswith(MAGIC):
            raise ValueError("not a rosbag 2.0 file: %s" % path)
        self.conns = {}

    def _walk(self, buf, p=0):
        for f, data in _records(buf, p):
            op = f["op"][0]
            if op == OP_CONNECTION:
                (cid,) = struct.unpack("<I", f["conn"])
                cf = _fields(data)
                self.conns[cid] = (f["topic"].decode(), cf.get("type", b"").decode())
            elif op == OP_CHUNK:
                comp = f["compression"].decode()
                if comp == "none":
                    inner = data
                elif comp == "bz2":
                    inner = bz2.decompress(data)
                else:
                    raise ValueError("chunk compression %r is not supported (re-record or `rosbag decompress` the bag)" % comp)
                yield from self._walk(inner)
            elif op == OP_MSG_DATA:
                (cid,) = struct.unpack("<I", f["conn"])
                secs, nsecs = struct.unpack("<II", f["time"])
                topic, typ = self.conns.get(cid, ("?", "?"))
                yield topic, typ, secs + nsecs * 1e-9, data

    def messages(self):
        return self._walk(self.buf, len(MAGIC))


class RosbagSequence:
    """Stereo pairs with EQUAL header stamps (what the reference's ExactTime synchroniser delivers, vo_tracking.cpp:308-319)
    and the IMU samples between consecutive pairs, from a bag with the reference's topic names (remappable)."""

    def __init__(self, path, img0_topic="/vo/input_image_0", img1_topic="/vo/input_image_1", imu_topic="/imu"):
        left, right, imu = {}, {}, []
        for topic, typ, _, data in BagReader(path).messages():
            if topic == img0_topic or topic == img1_topic:
                stamp, img, _ = parse_image(data)
                (left if topic == img0_topic else right)[int(round(stamp * 1e9))] = img
            elif topic == imu_topic:
                stamp, gyro, acc = parse_imu(data)
                imu.append(np.concatenate([[stamp], gyro, acc]))
        self.stamps_ns = sorted(set(left) & set(right))
        self.pairs = [(left[k], right[k]) for k in self.stamps_ns]
        self.imu = np.array(sorted(imu, key=lambda r: r[0])).reshape(-1, 7)   # t, gyro xyz, acc xyz (sensor frame)
        self.groundtruth = None

    def __len__(self):
        return len(self.stamps_ns)

    def frames(self, first=0, count=None):
        """yields (t_seconds, img0, img1, imu_rows) like traj_io.EurocSequence.frames"""
        last = len(self) if count is None else min(len(self), first + count)
        ti = self.imu[:, 0] if len(self.imu) else np.zeros(0)
        t_prev = -np.inf if first == 0 else self.stamps_ns[first - 1] * 1e-9
        for k in range(first, last):
            t = self.stamps_ns[k] * 1e-9
            sel = (ti > t_prev) & (ti <= t)
            i0, i1 = self.pairs[k]
            yield t, i0, i1, self.imu[sel]
            t_prev = t


# ------------------------------------------------------------------------------------------------- writer (test fixtures)
def _rec(fields, data):
    hdr = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k.encode() + b"=" + v for k, v in fields)
    return struct.pack("<I", len(hdr)) + hdr + struct.pack("<I", len(data)) + data


def _time(t):
    secs = int(np.floor(t))
    return struct.pack("<II", secs, int(round((t - secs) * 1e9)))


def ser_header(seq, t, frame_id):
    return struct.pack("<I", seq) + _time(t) + struct.pack("<I", len(frame_id)) + frame_id.encode()


def ser_image(seq, t, img, encoding="mono8"):
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    step = img.strides[0]
    raw = img.tobytes()
    return (ser_header(seq, t, "cam") + struct.pack("<II", h, w) + struct.pack("<I", len(encoding)) + encoding.encode() +
            struct.pack("<BI", 0, step) + struct.pack("<I", len(raw)) + raw)


def ser_imu(seq, t, gyro, acc):
    v = [0.0] * 37
    v[3] = 1.0
    v[13:16] = list(gyro)
    v[25:28] = list(acc)
    return ser_header(seq, t, "imu") + struct.pack("<37d", *v)


class BagWriter:
    """Writes an uncompressed single-chunk-per-flush rosbag 2.0 file with sensor_msgs/Image and sensor_msgs/Imu messages."""

    def __init__(self, path, compression="none"):
        self.path, self.compression = path, compression
        self.conn_ids, self.chunk = {}, b""
        self.out = [MAGIC, _rec([("op", bytes([OP_BAG_HEADER])), ("index_pos", struct.pack("<Q", 0)), ("conn_count", struct.pack("<I", 0)),
                                 ("chunk_count", struct.pack("<I", 0))], b" " * 64)]

    def _conn(self, topic, typ):
        if topic not in self.conn_ids:
            cid = len(self.conn_ids)
            self.conn_ids[topic] = cid
            inner = b"".join(struct.pack("<I", len(k) + 1 + len(v)) + k.encode() + b"=" + v
                             for k, v in (("topic", topic.encode()), ("type", typ.encode()), ("md5sum", b"0" * 32), ("message_definition", b"")))
            self.chunk += _rec([("op", bytes([OP_CONNECTION])), ("conn", struct.pack("<I", cid)), ("topic", topic.encode())], inner)
        return self.conn_ids[topic]

    def write(self, topic, typ, t, payload):
        cid = self._conn(topic, typ)
        self.chunk += _rec([("op", bytes([OP_MSG_DATA])), ("conn", struct.pack("<I", cid)), ("time", _time(t))], payload)
        if len(self.chunk) > (1 << 20):
            self.flush()

    def flush(self):
        if not self.chunk:
            return
        data = self.chunk if self.compression == "none" else bz2.compress(self.chunk)
        self.out.append(_rec([("op", bytes([OP_CHUNK])), ("compression", self.compression.encode()), ("size", struct.pack("<I", len(self.chunk)))], data))
        self.chunk = b""

    def close(self):
        self.flush()
        with open(self.path, "wb") as f:
            f.write(b"".join(self.out))
