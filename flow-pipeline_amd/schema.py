"""FlowMessage schema tables (host side).

A restatement, as data, of the two schemas the reference carries for the same
message:

* ``LIGHT``  - the 27-field ``pb-ext/flow.proto:7-65``.  This is the file
  ClickHouse reads (``kafka_schema='flow.proto:FlowMessage'``,
  ``compose/clickhouse/create.sh:33-34``), i.e. the schema of the hot path.
* ``FULL``   - the 67-field Go struct ``pb-ext/flow.pb.go:57-147`` (superset:
  MACs, VLANs, VRF, encap, MPLS, PPP, countries ...) that the mocker / GoFlow
  producers marshal with.

Only the wire *kind* matters to the decoder: ``v`` = varint scalar
(uint32/uint64/bool/enum), ``b`` = length-delimited bytes/string.

``PROJECTED`` is the ClickHouse ``flows`` column list
(``compose/clickhouse/create.sh:7-27``): column name -> (proto field number,
kind, column type).  Note the column is ``EType`` while the proto field is
``Etype`` (``flow.proto:40``).

The module can also build a protobuf (upb) message class from these tables with
no ``protoc`` - used by the Python mocker mirror and by the tests as an
independent encoder/decoder.  That part needs the ``protobuf`` wheel; the
tables themselves do not.
"""
from __future__ import annotations

# (name, number, kind, proto type)
LIGHT = [
    ("Type", 1, "v", "enum"),
    ("TimeReceived", 2, "v", "uint64"),
    ("SamplingRate", 3, "v", "uint64"),
    ("SequenceNum", 4, "v", "uint32"),
    ("TimeFlowEnd", 5, "v", "uint64"),
    ("SrcAddr", 6, "b", "bytes"),
    ("DstAddr", 7, "b", "bytes"),
    ("Bytes", 9, "v", "uint64"),
    ("Packets", 10, "v", "uint64"),
    ("SamplerAddress", 11, "b", "bytes"),
    ("SrcAS", 14, "v", "uint32"),
    ("DstAS", 15, "v", "uint32"),
    ("InIf", 18, "v", "uint32"),
    ("OutIf", 19, "v", "uint32"),
    ("Proto", 20, "v", "uint32"),
    ("SrcPort", 21, "v", "uint32"),
    ("DstPort", 22, "v", "uint32"),
    ("IPTos", 23, "v", "uint32"),
    ("ForwardingStatus", 24, "v", "uint32"),
    ("IPTTL", 25, "v", "uint32"),
    ("TCPFlags", 26, "v", "uint32"),
    ("Etype", 30, "v", "uint32"),
    ("IcmpType", 31, "v", "uint32"),
    ("IcmpCode", 32, "v", "uint32"),
    ("IPv6FlowLabel", 37, "v", "uint32"),
    ("TimeFlowStart", 38, "v", "uint64"),
    ("FlowDirection", 42, "v", "uint32"),
]

_FULL_EXTRA = [
    ("NextHop", 12, "b", "bytes"),
    ("NextHopAS", 13, "v", "uint32"),
    ("SrcNet", 16, "v", "uint32"),
    ("DstNet", 17, "v", "uint32"),
    ("SrcMac", 27, "v", "uint64"),
    ("DstMac", 28, "v", "uint64"),
    ("VlanId", 29, "v", "uint32"),
    ("SrcVlan", 33, "v", "uint32"),
    ("DstVlan", 34, "v", "uint32"),
    ("FragmentId", 35, "v", "uint32"),
    ("FragmentOffset", 36, "v", "uint32"),
    ("IngressVrfID", 39, "v", "uint32"),
    ("EgressVrfID", 40, "v", "uint32"),
    ("BiFlowDirection", 41, "v", "uint32"),
    ("HasEncap", 43, "v", "bool"),
    ("SrcAddrEncap", 44, "b", "bytes"),
    ("DstAddrEncap", 45, "b", "bytes"),
    ("ProtoEncap", 46, "v", "uint32"),
    ("EtypeEncap", 47, "v", "uint32"),
    ("IPTosEncap", 48, "v", "uint32"),
    ("IPTTLEncap", 49, "v", "uint32"),
    ("IPv6FlowLabelEncap", 50, "v", "uint32"),
    ("FragmentIdEncap", 51, "v", "uint32"),
    ("FragmentOffsetEncap", 52, "v", "uint32"),
    ("HasMPLS", 53, "v", "bool"),
    ("MPLSCount", 54, "v", "uint32"),
    ("MPLS1TTL", 55, "v", "uint32"),
    ("MPLS1Label", 56, "v", "uint32"),
    ("MPLS2TTL", 57, "v", "uint32"),
    ("MPLS2Label", 58, "v", "uint32"),
    ("MPLS3TTL", 59, "v", "uint32"),
    ("MPLS3Label", 60, "v", "uint32"),
    ("MPLSLastTTL", 61, "v", "uint32"),
    ("MPLSLastLabel", 62, "v", "uint32"),
    ("HasPPP", 63, "v", "bool"),
    ("PPPAddressControl", 64, "v", "uint32"),
    ("SrcCountry", 100, "b", "string"),
    ("DstCountry", 101, "b", "string"),
    ("SrcASDB", 102, "v", "uint32"),
    ("DstASDB", 103, "v", "uint32"),
]
FULL = sorted(LIGHT + _FULL_EXTRA, key=lambda f: f[1])

# ClickHouse `flows` columns (create.sh:7-27), in DDL order:
# column -> (proto field number, kind, column type)
PROJECTED = {
    "TimeReceived": (2, "v", "UInt64"),
    "TimeFlowStart": (38, "v", "UInt64"),
    "SequenceNum": (4, "v", "UInt32"),
    "SamplingRate": (3, "v", "UInt64"),
    "SamplerAddress": (11, "b", "FixedString(16)"),
    "SrcAddr": (6, "b", "FixedString(16)"),
    "DstAddr": (7, "b", "FixedString(16)"),
    "SrcAS": (14, "v", "UInt32"),
    "DstAS": (15, "v", "UInt32"),
    "EType": (30, "v", "UInt32"),
    "Proto": (20, "v", "UInt32"),
    "SrcPort": (21, "v", "UInt32"),
    "DstPort": (22, "v", "UInt32"),
    "Bytes": (9, "v", "UInt64"),
    "Packets": (10, "v", "UInt64"),
}

FLOW_TYPES = ["FLOWUNKNOWN", "SFLOW_5", "NETFLOW_V5", "NETFLOW_V9", "IPFIX"]

_PB_TYPE = {"uint64": 4, "uint32": 13, "bool": 8, "bytes": 12, "string": 9, "enum": 14}


def file_descriptor_proto(fields=None, package="flowprotob", fname="flow.proto"):
    """Build a FileDescriptorProto for FlowMessage from a field table."""
    from google.protobuf import descriptor_pb2

    fields = LIGHT if fields is None else fields
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.name = fname
    fdp.package = package
    fdp.syntax = "proto3"
    msg = fdp.message_type.add()
    msg.name = "FlowMessage"
    en = msg.enum_type.add()
    en.name = "FlowType"
    for i, n in enumerate(FLOW_TYPES):
        v = en.value.add()
        v.name = n
        v.number = i
    for name, number, _kind, ptype in fields:
        f = msg.field.add()
        f.name = name
        f.number = number
        f.label = 1  # optional
        f.type = _PB_TYPE[ptype]
        f.json_name = name
        if ptype == "enum":
            f.type_name = ".%s.FlowMessage.FlowType" % package
    return fdp


_CLASS_CACHE = {}


def message_class(which="light"):
    """Return a upb-backed FlowMessage class for the LIGHT or FULL schema."""
    if which in _CLASS_CACHE:
        return _CLASS_CACHE[which]
    from google.protobuf import descriptor_pool, message_factory

    fields = LIGHT if which == "light" else FULL
    pool = descriptor_pool.DescriptorPool()
    # distinct file/package names so both schemas can coexist in one process
    fdp = file_descriptor_proto(fields, package="flowprotob_%s" % which,
                                fname="flow_%s.proto" % which)
    fd = pool.Add(fdp)
    cls = message_factory.GetMessageClass(fd.message_types_by_name["FlowMessage"])
    _CLASS_CACHE[which] = cls
    return cls


def encode_varint(v: int) -> bytes:
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def frame(payload: bytes) -> bytes:
    """``proto.Buffer.EncodeMessage`` framing: varint(len) || payload
    (``mocker/mocker.go:98-101``, ``-proto.fixedlen=true``)."""
    return encode_varint(len(payload)) + payload
