"""MI355X-native PerspectiveFields dense-field + ParamNet inference path."""
