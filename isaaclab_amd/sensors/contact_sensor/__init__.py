from .contact_sensor import ContactSensor, ContactSensorCfg, ContactSensorData
