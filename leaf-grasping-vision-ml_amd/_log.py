"""rospy-compatible logging shim: the reference logs through rospy (absent here); use rospy when it is
importable, python logging otherwise (the reference's demos use the same trick,
vla_system/demos/test_vla_simple.py:10-15)."""
import logging

try:  # pragma: no cover - rospy is not in this image
    import rospy as _rospy

    loginfo, logwarn, logerr, logdebug = _rospy.loginfo, _rospy.logwarn, _rospy.logerr, _rospy.logdebug
except Exception:  # noqa: BLE001
    _l = logging.getLogger("leafgrasp_amd")
    loginfo, logwarn, logerr, logdebug = _l.info, _l.warning, _l.error, _l.debug
