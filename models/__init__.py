"""Drop-in `models` package: same module / class names as kigb/DropoutDecoding's models/, MI355X-native underneath."""
